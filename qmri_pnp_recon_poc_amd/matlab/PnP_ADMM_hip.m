function [x, diag, lsqr_iters] = PnP_ADMM_hip(y, param)
% PNP_ADMM_HIP  Drop-in for  x = PnP_ADMM(y, param)  (main_files/algorithms/PnP_ADMM/PnP_ADMM.m:1) that runs the whole
%   loop on the GPU (one boundary crossing per reconstruction).  param.F must come from qmri_make_F and param.net from
%   qmri_make_net; the fields read are the reference's own: iter, gamma, cg_tol, gt_tsmi, X0, denoiser_type, noise_map
%   (PnP_ADMM.m:62-76).  Extra outputs: the two per-iteration diagnostics (PnP_ADMM.m:106-109) and the LSQR iteration counts.
if ~isfield(param.F, 'qmri'), error('qmri:F', 'param.F must be created by qmri_make_F'); end
p.gamma = param.gamma;  p.iter = param.iter;  p.cg_tol = param.cg_tol;
p.multi_level = double(strcmp(param.denoiser_type, 'multi_level'));
if p.multi_level, p.noise_std = param.noise_map(1); else, p.noise_std = 0.01; end
g = param.F.qmri;
gt = [];  if isfield(param, 'gt_tsmi'), gt = complex(double(param.gt_tsmi)); end
X0 = [];  if isfield(param, 'X0'), X0 = complex(double(param.X0)); end
if nargout > 1
    [x, diag, lsqr_iters] = qmri_mex('pnp_admm', complex(double(y(:))), p, X0, gt, [g.N g.M g.s]);
    diag = diag.';
else
    x = qmri_mex('pnp_admm', complex(double(y(:))), p, X0, gt, [g.N g.M g.s]);
end
end
