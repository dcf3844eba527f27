function [nlogML,grad,w,iSigma_w,PHI] = GPz(theta,model,X,Y,Psi,omega,training,validation)
% Drop-in replacement of GPz/GPz.m that forwards to libgpz_hip.so through gpz_mex.
% Same signature, outputs and side effects (globals) as the reference file it replaces.
%
% The device context of the closure (train.m:40) lives inside gpz_mex between calls.  Every call hands the closure's
% arguments over again (shared-data copies: no cost); the gateway rebuilds the context when the model fields, any
% array's size, data pointer or sampled content differ from what the live context was built from.  gpz_mex('reset')
% drops the context explicitly (e.g. after edits MATLAB may have done in place).
% model.n_gpus (optional field): number of GPUs, default all GPUs of the node.

global trainRMSE
global trainLL
global validRMSE
global validLL

if(isempty(Y))                                  % GPz.m:34-40
    nlogML = 0; grad = 0; w = 0; iSigma_w = 0;
    return
end

if(nargout>2)                                   % GPz.m:84-87: solve only, globals untouched
    [w,iSigma_w,nlogML] = gpz_mex('solve',theta,model,X,Y,Psi,omega,training,validation);
    grad = 0;
    if(nargout>4)
        PHI = gpz_mex('phi');
    end
    return
end

[nlogML,grad,stats] = gpz_mex('eval',theta,model,X,Y,Psi,omega,training,validation);
trainRMSE = stats(1);                           % GPz.m:236-237
trainLL   = stats(2);
if(~isempty(validation))
    validRMSE = stats(3);                       % GPz.m:258-259
    validLL   = stats(4);
end

end
