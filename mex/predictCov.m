function [mu,nu,beta_i,gamma,PHI] = predictCov(X,Psi,model,set,ind)
% Drop-in replacement of GPz/predictCov.m (called per NaN-pattern group by predict.m:60-69) that forwards to
% libgpz_hip.so through gpz_mex: same signature and outputs.  Psi is the d x d x n cube predict.m hands over
% (predict.m:28-29,43); the gateway picks the branch of predictCov.m:36-52 from what X(ind,:) and Psi(:,:,ind) contain.

if(isempty(Psi))
    Psi_g = [];
else
    Psi_g = Psi(:,:,ind);
end
if(nargout>4)
    [mu,nu,beta_i,gamma,PHI] = gpz_mex('predict',model,set.theta,set.w,set.iSigma_w,set.priors,X(ind,:),Psi_g);
else
    [mu,nu,beta_i,gamma] = gpz_mex('predict',model,set.theta,set.w,set.iSigma_w,set.priors,X(ind,:),Psi_g);
end

end
