function F = qmri_make_F(pattern, N, M, rate, V)
% QMRI_MAKE_F  GPU forward/adjoint operator with the reference's plugin surface.
%   Replaces main_recon_tsmis_FFT.m:220-229:
%       [P] = setup_subsampling_spiralgrided(N,M,spiral_sampling_curve,V);   (or setup_subsampling_epi)
%       F.forward = @(x) P.for(reshape(fft2(x),[],1))/sqrt(N*M);
%       F.adjoint = @(x) (ifft2(reshape(P.adj(x),N,M,[]))*sqrt(N*M));
%   by   F = qmri_make_F('Spiral', N, M, spiral_sampling_curve, V);          (or 'EPI', ..., epi_sampling_rate, V)
T = size(V, 1);  s = size(V, 2);
switch pattern
    case 'Spiral', [fp, k] = qmri_mex('build_spiral', N, rate, T);
    case 'EPI',    [fp, k] = qmri_mex('build_epi', N, M, rate, T);
    otherwise, error('qmri:pattern', 'unknown subsampling pattern %s', pattern);
end
qmri_mex('set_operator', N, M, real(double(V)), fp, k);
F.forward = @(x) qmri_mex('forward', double(x));
F.adjoint = @(y) qmri_mex('adjoint', complex(double(y)), [N M s]);
F.qmri = struct('N', N, 'M', M, 's', s);      % marks F as GPU resident for PnP_ADMM_hip
end
