function prior = getPrior(X,Sx,theta,model,set)
% Drop-in replacement of GPz/getPrior.m that forwards to libgpz_hip.so through gpz_mex: same signature and output.
% N (getPHI.m:114) does not depend on the prior, so the device computes it once and iterates the fixed point of
% getPrior.m:7-20 on it instead of rebuilding PHI up to 100 times.

if(~isempty(set))                               % getPHI.m:3-22 does this selection inside the reference's loop
    X = X(set,:);
    if(~isempty(Sx))
        if(model.method(2)=='C'), Sx = Sx(:,:,set); else, Sx = Sx(set,:); end
    end
end
prior = gpz_mex('prior',model,theta,X,Sx);

end
