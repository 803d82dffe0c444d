function [X, qmap, pd] = qmri_recon_batch(Y, param, devs, slices_per_launch)
% QMRI_RECON_BATCH  A whole slice stack through PnP-ADMM (+ the dictionary match) on the GPUs of one node.
%   The reference reconstructs one slice per run of main_recon_tsmis_FFT.m (:37-38, :293, :317); slices are independent, so a stack of S
%   slices is sharded over the devices in `devs` -- one worker (host thread + context) per entry, no communication between them -- and
%   on each device slices_per_launch slices advance together through the batched kernels:
%
%       F         = qmri_make_F('Spiral', N, M, spiral_sampling_curve, V);
%       param.net = qmri_make_net(denoiser_path, param.denoiser_type, residual_noise);
%       mrf_dtm_hip(dict, [], []);                        % (or any earlier call: leaves the dictionary set; optional)
%       [X, qmap, pd] = qmri_recon_batch(Y, param, 0:7, 15);     % Y: m x S, column j = F.forward(X0_j) + noise
%
%   X: N x M x s x S complex double (PnP_ADMM's x of every slice); qmap: N x M x Q x S single and pd: N x M x S complex single (mrf_dtm_cpu's
%   out.qmap / out.pd of every slice; empty when no dictionary is set or they are not asked for).
%   devs (default 0): device ids, an id may repeat (two workers sharing one GPU); slices_per_launch (default 15).
%   param: the reference's fields iter, gamma, cg_tol, denoiser_type, noise_map (PnP_ADMM.m:62-76); param.F from qmri_make_F.
if nargin < 3 || isempty(devs), devs = 0; end
if nargin < 4 || isempty(slices_per_launch), slices_per_launch = 15; end
if ~isfield(param.F, 'qmri'), error('qmri:F', 'param.F must be created by qmri_make_F'); end
p.gamma = param.gamma;  p.iter = param.iter;  p.cg_tol = param.cg_tol;
p.multi_level = double(strcmp(param.denoiser_type, 'multi_level'));
if p.multi_level, p.noise_std = param.noise_map(1); else, p.noise_std = 0.01; end
g = param.F.qmri;
if nargout > 1
    [X, qmap, pd] = qmri_mex('recon_batch', complex(double(Y)), p, double(devs(:)), double(slices_per_launch), [g.N g.M g.s]);
else
    X = qmri_mex('recon_batch', complex(double(Y)), p, double(devs(:)), double(slices_per_launch), [g.N g.M g.s]);
end
end
