function [nlogML,grad,w,iSigma_w,PHI] = GPz(theta,model,X,Y,Psi,omega,training,validation)
% Drop-in replacement of GPz/GPz.m that forwards to libgpz_hip.so through gpz_mex.
% Same signature, outputs and side effects (globals) as the reference file it replaces.

global trainRMSE
global trainLL
global validRMSE
global validLL

persistent key
if(isempty(Y))                                  % GPz.m:34-40
    nlogML = 0; grad = 0; w = 0; iSigma_w = 0;
    return
end

% one device context per closure: re-create when the captured data changes
newkey = [size(X) size(Y) numel(Psi) numel(omega) sum(training(:)) sum(validation(:)) X(1) X(end) Y(1) Y(end)];
if(~isequal(key,newkey))
    gpz_mex('create',model,X,Y,Psi,omega,training,validation);
    key = newkey;
end

if(nargout>2)                                   % GPz.m:84-87: solve only, globals untouched
    [w,iSigma_w,nlogML] = gpz_mex('solve',theta);
    grad = 0;
    if(nargout>4)
        PHI = gpz_mex('phi');
    end
    return
end

[nlogML,grad,stats] = gpz_mex('eval',theta);
trainRMSE = stats(1);                           % GPz.m:236-237
trainLL   = stats(2);
if(~isempty(validation))
    validRMSE = stats(3);                       % GPz.m:258-259
    validLL   = stats(4);
end

end
