function [mu,nu,beta_i,gamma,PHI] = predictDiag(X,Psi,model,set,ind)
% Drop-in replacement of GPz/predictDiag.m (called per NaN-pattern group by predict.m:60-69) that forwards to
% libgpz_hip.so through gpz_mex: same signature and outputs.  The gateway picks predictFull / predictNoisy /
% predictMissing / predictNoisyMissing from what X(ind,:) and Psi(ind,:) contain, as predictDiag.m:39-55 does.

if(isempty(Psi))
    Psi_g = [];
else
    Psi_g = Psi(ind,:);
end
if(nargout>4)
    [mu,nu,beta_i,gamma,PHI] = gpz_mex('predict',model,set.theta,set.w,set.iSigma_w,set.priors,X(ind,:),Psi_g);
else
    [mu,nu,beta_i,gamma] = gpz_mex('predict',model,set.theta,set.w,set.iSigma_w,set.priors,X(ind,:),Psi_g);
end

end
