function net = qmri_make_net(weights, denoiser_type, residual_noise)
% QMRI_MAKE_NET  GPU denoiser handle with the signature of param.net (main_recon_tsmis_FFT.m:164):
%       Net = importONNXNetwork(denoiser_path, 'OutputLayerType', 'regression');                     (:138)
%       param.net = @(x) denoiseImage_PnP_ADMM(x, Net, onnx_dagnetwork, residual_noise);             (:164)
%   becomes   param.net = qmri_make_net(denoiser_path, param.denoiser_type, residual_noise);
%   weights: either the path of the ONNX file export_to_onnx wrote (PyTorch_Denoiser/utils.py:468-481; read by the
%   library itself, no Deep Learning Toolbox involved), or a single vector in UNetRes state_dict order (Conv2d OIHW,
%   ConvTranspose2d IOHW): 32 648 448 values for the 10-channel and 32 649 024 for the 11-channel DRUNet
%   (main_test.py:245-252).
switch denoiser_type
    case 'single_level', in_nc = 10;
    case 'multi_level',  in_nc = 11;
    otherwise, error('qmri:denoiser_type', 'unknown denoiser type %s', denoiser_type);
end
if ischar(weights) || isstring(weights)
    got = qmri_mex('load_onnx', char(weights), double(residual_noise), 224, 224);
    if got ~= in_nc
        error('qmri:denoiser_type', '%s takes %d input channels, denoiser type %s needs %d', char(weights), got, denoiser_type, in_nc);
    end
else
    qmri_mex('set_denoiser', single(weights(:)), in_nc, 10, [64 128 256 512], 4, double(residual_noise), 224, 224);
end
net = @(x) qmri_mex('denoise', double(x), 10);
end
