import sys, numpy as np, torch
sys.path.insert(0, ".")
from sdr_channelizer_amd import Channelizer, design_prototype, synth
from sdr_channelizer_amd import _lib as L
n = 1 << 30
dev = torch.device("cuda", 0)
iq = synth.pulsed_iq_torch(n, 12, torch.int16, device=dev)
out = torch.empty((n // 64, 64), dtype=torch.complex64, device=dev)
ch = Channelizer(64, taps=design_prototype(64, 12), bit_width=12)
ch.set_stream(torch.cuda.current_stream(dev).cuda_stream)
ch.set_option(L.PFB_OPT_KERNEL, 2)
ch.set_option(L.PFB_OPT_PROFILE, 1)
torch.cuda.synchronize()
for _ in range(60):
    ch(iq, out=out, sync=False)
t = np.array(ch.kernel_times_ms())
print("launch times ms:", np.round(t, 3).tolist())
print("mean first 23:", t[:23].mean(), "mean 3..23:", t[3:23].mean(), "mean last 20:", t[-20:].mean())
