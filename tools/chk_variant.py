import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from sdr_channelizer_amd import Channelizer, design_prototype, synth
from sdr_channelizer_amd import _lib as L
M, P = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 16)
n = M * 4000 + 5
dev=torch.device('cuda',0)
iq=synth.pulsed_iq_torch(n,12,torch.int16,device=dev)
ch=Channelizer(M,taps=design_prototype(M,P),decimation=M,sample_format='int16',bit_width=12)
ch.set_option(L.PFB_OPT_KERNEL,1)
g=ch(iq).clone(); s=g.abs().max().item()
ch.set_option(L.PFB_OPT_KERNEL,2)
for v in range(8):
    try:
        ch.set_option(L.PFB_OPT_VARIANT,v)
    except Exception as e:
        break
    ch.reset()
    for fl in (0,):
        a=ch(iq).clone()
        print(v, ch.last_kernel, 'max err vs generic / max|y| =', (a-g).abs().max().item()/s)
