function [x, diag, lsqr_iters] = PnP_ADMM_hip(y, param)
% PNP_ADMM_HIP  Drop-in for  x = PnP_ADMM(y, param)  (main_files/algorithms/PnP_ADMM/PnP_ADMM.m:1) that runs the whole
%   loop on the GPU (one boundary crossing per reconstruction).  param.F must come from qmri_make_F and param.net from
%   qmri_make_net; the fields read are the reference's own: iter, gamma, cg_tol, gt_tsmi, X0, denoiser_type, noise_map
%   (PnP_ADMM.m:62-76).  Extra outputs: the two per-iteration diagnostics (PnP_ADMM.m:106-109) and the LSQR iteration counts.
%
%   y is the measurement vector of one slice (m x 1, as in the reference) or a measurement MATRIX m x S, one column per slice:
%   the S slices then advance together through the batched kernels (15 at a time) on the current device and x is
%   N x M x s x S (diag: S x iter x 2 -> returned as iter x 2 x S; lsqr_iters: iter x S); param.X0 / param.gt_tsmi, if given, are
%   N x M x s x S.  For S slices over SEVERAL GPUs see qmri_recon_batch.
if ~isfield(param.F, 'qmri'), error('qmri:F', 'param.F must be created by qmri_make_F'); end
p.gamma = param.gamma;  p.iter = param.iter;  p.cg_tol = param.cg_tol;
p.multi_level = double(strcmp(param.denoiser_type, 'multi_level'));
if p.multi_level, p.noise_std = param.noise_map(1); else, p.noise_std = 0.01; end
g = param.F.qmri;
if isvector(y), y = y(:); end
gt = [];  if isfield(param, 'gt_tsmi'), gt = complex(double(param.gt_tsmi)); end
X0 = [];  if isfield(param, 'X0'), X0 = complex(double(param.X0)); end
if nargout > 1
    [x, diag, lsqr_iters] = qmri_mex('pnp_admm', complex(double(y)), p, X0, gt, [g.N g.M g.s]);
    diag = permute(diag, [2 1 3]);
else
    x = qmri_mex('pnp_admm', complex(double(y)), p, X0, gt, [g.N g.M g.s]);
end
end
