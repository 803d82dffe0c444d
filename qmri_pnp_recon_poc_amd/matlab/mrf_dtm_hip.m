function out = mrf_dtm_hip(dict, data, par)
% MRF_DTM_HIP  Drop-in for  out = mrf_dtm_cpu(dict, data, par)  (main_files/dictionary_matching/mrf_dtm_cpu.m:1).
%   Same fields in, same fields out (qmap, pd, mt, dm, mask, X as gated by par.f.*); par.fp.blockSize is accepted and
%   ignored (the K x Npix product is never materialised on the GPU).
if ~isempty(data), datadims = size(data.X); T = datadims(end); Npix = prod(datadims(1:end-1)); end
Q = size(dict.lut, 2);
D = dict.D;
if ~isreal(D)   % mrf_dtm_cpu.m:91 multiplies by dict.D as stored; the GPU match implements real atoms
    if any(imag(D(:)) ~= 0)
        error('qmri:mrf_dtm_hip:complexDictionary', 'dict.D has a non-zero imaginary part; the GPU dictionary match takes real atoms (use mrf_dtm_cpu, or pass real(dict.D) if that is what is meant)');
    end
    D = real(D);
end
qmri_mex('set_dictionary', single(D), single(real(dict.normD(:))), single(real(dict.lut)));
if isempty(data), out = []; return; end     % mrf_dtm_hip(dict, [], []): only leave the dictionary set (qmri_recon_batch uses it)
if par.f.Xout   % mrf_dtm_cpu.m:95,129-134: Xfit = ip(dm) .* D(dm,:)
    [qmap, pd, mt, dm, xfit] = qmri_mex('dict_match', complex(double(reshape(data.X, [Npix, T]))), Q);
else
    [qmap, pd, mt, dm] = qmri_mex('dict_match', complex(double(reshape(data.X, [Npix, T]))), Q);
end
if par.f.qout,  out.qmap = reshape(qmap, [datadims(1:end-1), Q]);  out.mask = true(datadims(1:end-1)); end
if par.f.pdout, out.pd = reshape(pd, [datadims(1:end-1), 1]); end
if par.f.mtout, out.mt = reshape(mt, [datadims(1:end-1), 1]); end
if par.f.dmout, out.dm = reshape(single(dm), [datadims(1:end-1), 1]); end
if par.f.Xout,  out.Xfit = reshape(xfit, [datadims(1:end-1), T]);  out.X = data.X; end
if par.f.Yout && isfield(data, 'Y'), out.Y = data.Y; end
end
