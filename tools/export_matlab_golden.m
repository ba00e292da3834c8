function export_matlab_golden(cfg)
% EXPORT_MATLAB_GOLDEN  One-file pin of the channelizer arithmetic against MATLAB's dsp.Channelizer.
%   Run from the repository root in MATLAB with the DSP System Toolbox:
%       export_matlab_golden('cfg2')        % or cfg1 / cfg3 / cfg4 / cfg5 / ref56 / ref560
%   reads  tests/golden/matlab_in_<cfg>.mat   (fixed integer input, written by tests/golden/make_matlab_inputs.py)
%   writes tests/golden/matlab_<cfg>.mat      (taps, output, centre frequencies, version)
%   Drop the output file into tests/golden/ and run  pytest tests/test_matlab_pin.py : the test searches the
%   alignment switches (input_offset, derotate, conjugate, output scale) and asserts <= 1e-5 (INTEGRATION.md section 6).
% Follows the reference's own call sequence: matlab/channelizer_example.m:18-31,56,60.
in = load(fullfile('tests', 'golden', ['matlab_in_' cfg '.mat']));
M = double(in.M); D = double(in.D); P = double(in.P); fs = double(in.fs);
if in.is_float                                                        % cfg1: cfloat I/Q, already normalised
    x = double(in.iq(:, 1)) + 1j * double(in.iq(:, 2));
else
    x = (double(in.iq(:, 1)) + 1j * double(in.iq(:, 2))) / 2^(double(in.bit_width) - 1);   % :18-21
end
x = x(1:floor(numel(x) / M) * M);                                     % create_pdws_channelized.m:52-54
channelizer = dsp.Channelizer(M);                                     % :31, every other property at its default
if P ~= 12, channelizer.NumTapsPerBand = P; end                       % BASELINE configs with 8 / 16 taps per band
if D ~= M,  channelizer.DecimationFactor = D; end                     % cfg5: 2x oversampled
out = channelizer(x);                                                 % :56  (F x M, complex double)
c = coeffs(channelizer);                                              % the prototype low-pass MATLAB designed
if isstruct(c), taps = c.Numerator; else, taps = c; end
polyphase_matrix = polyphase(channelizer);                            %#ok<NASGU> M x taps-per-branch, as MATLAB splits it
center_frequencies = centerFrequencies(channelizer, fs);              %#ok<NASGU> :60 -- settles the order question
matlab_version = version;                                             %#ok<NASGU>
toolbox = ver('dsp');                                                 %#ok<NASGU>
taps = double(taps(:).');                                             %#ok<NASGU>
save(fullfile('tests', 'golden', ['matlab_' cfg '.mat']), 'out', 'taps', 'polyphase_matrix', 'center_frequencies', ...
     'matlab_version', 'toolbox', 'M', 'P', 'D', 'fs', '-v7');
fprintf('wrote tests/golden/matlab_%s.mat: out %dx%d, %d taps\n', cfg, size(out, 1), size(out, 2), numel(taps));
end
