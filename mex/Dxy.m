function D = Dxy(X,Y)
% Drop-in replacement of GPz/Dxy.m (callers init.m:62, getOmega.m:16) that forwards to libgpz_hip.so through gpz_mex.
D = gpz_mex('dxy',X,Y);
end
