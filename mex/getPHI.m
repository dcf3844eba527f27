function [PHI,Gamma,lnBeta_i,N] = getPHI(X,Psi,theta,model,selection)
% Drop-in replacement of GPz/getPHI.m that forwards to libgpz_hip.so through gpz_mex (same signature and outputs).

if(~isempty(selection))                         % getPHI.m:3-22: row selection stays here
    X = X(selection,:);
    if(~isempty(Psi))
        if(model.method(2)=='C'), Psi = Psi(:,:,selection); else, Psi = Psi(selection,:); end
    end
end

if(nargout>3)
    [PHI,lnBeta_i,N] = gpz_mex('getphi',model,theta,X,Psi);
else
    [PHI,lnBeta_i] = gpz_mex('getphi',model,theta,X,Psi);
end

m = model.m; d = model.d;                       % getPHI.m:28-39: Gamma is a reshape of theta
switch(model.method)
    case 'GL', Gamma = repmat(theta(m*d+1),m,d);
    case 'VL', Gamma = repmat(theta(m*d+1:m*d+m),1,d);
    case 'GD', Gamma = repmat(theta(m*d+1:m*d+d)',m,1);
    case 'VD', Gamma = reshape(theta(m*d+1:m*d+m*d),m,d);
    case 'GC', Gamma = repmat(reshape(theta(m*d+1:m*d+d*d),d,d),1,1,m);
    case 'VC', Gamma = reshape(theta(m*d+1:m*d+d*d*m),d,d,m);
end

end
