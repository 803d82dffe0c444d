function net = qmri_make_net(weights, denoiser_type, residual_noise, imsize, arch)
% QMRI_MAKE_NET  GPU denoiser handle with the signature of param.net (main_recon_tsmis_FFT.m:164):
%       Net = importONNXNetwork(denoiser_path, 'OutputLayerType', 'regression');                     (:138)
%       param.net = @(x) denoiseImage_PnP_ADMM(x, Net, onnx_dagnetwork, residual_noise);             (:164)
%   becomes   param.net = qmri_make_net(denoiser_path, param.denoiser_type, residual_noise);
%   weights: either the path of the ONNX file export_to_onnx wrote (PyTorch_Denoiser/utils.py:468-481; read by the
%   library itself, no Deep Learning Toolbox involved -- the channel counts then come from the file), or a single vector in
%   UNetRes state_dict order (Conv2d OIHW, ConvTranspose2d IOHW): 32 648 448 values for the 10-channel and 32 649 024 for
%   the 11-channel DRUNet (main_test.py:245-252).
%   imsize (optional, default [224 224], the script's cropped TSMI, :189,:212): [H W] of the images the handle will see.
%   arch   (optional, only with a weight vector): struct('out_nc', 10, 'nc', [64 128 256 512], 'nb', 4) = UNetRes as trained
%          (network_unet.py:68); the input has out_nc channels ('single_level') or out_nc + 1 ('multi_level': the noise map).
if nargin < 4 || isempty(imsize), imsize = [224 224]; end
if nargin < 5 || isempty(arch), arch = struct('out_nc', 10, 'nc', [64 128 256 512], 'nb', 4); end
switch denoiser_type
    case 'single_level', extra = 0;
    case 'multi_level',  extra = 1;
    otherwise, error('qmri:denoiser_type', 'unknown denoiser type %s', denoiser_type);
end
if ischar(weights) || isstring(weights)
    [got, out_nc] = qmri_mex('load_onnx', char(weights), double(residual_noise), imsize(1), imsize(2));
    if got ~= out_nc + extra
        error('qmri:denoiser_type', '%s takes %d input channels for %d output channels, denoiser type %s needs %d', char(weights), got, out_nc, denoiser_type, out_nc + extra);
    end
else
    out_nc = arch.out_nc;
    qmri_mex('set_denoiser', single(weights(:)), out_nc + extra, out_nc, double(arch.nc(:).'), arch.nb, double(residual_noise), imsize(1), imsize(2));
end
% The handle takes H x W x C or H x W x C x N, as the reference's does (denoiseImage_PnP_ADMM.m:13-17): a batch larger than the current
% plan makes the gateway re-plan for it (one-off cost of a weight re-pack), no argument needed here.
net = @(x) qmri_mex('denoise', double(x), out_nc);
end
