function [x, info] = FISTA_deep_hip(data, param)
% FISTA_DEEP_HIP  Drop-in for  [out.X] = FISTA_deep(data, param)  (main_recon_tsmis_FFT.m:282, FISTA_deep.m:1):
%   the whole LRTV loop -- FISTA with backtracking, unlocbox prox_tv on the stacked real/imaginary image -- runs on the GPU.
%   data.y  k-space measurements;  data.N, data.M, data.L  image dimensions;  data.F  must come from qmri_make_F (the
%   operator lives in the library);  param.K, .iter, .step, .tol, .backtrack as set at main_recon_tsmis_FFT.m:274-279
%   (param.usegpu and paramTV are not needed).
[x, info] = qmri_mex('lrtv', complex(double(data.y(:))), param, [data.M data.M data.L]);   % (the script passes data.N = M, :281)
end
