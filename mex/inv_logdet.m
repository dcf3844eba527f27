function [Xi,logdet] = inv_logdet(X)
% Drop-in replacement of GPz/inv_logdet.m (Cholesky inverse, or the truncating SVD route when X is numerically singular).
[Xi,logdet] = gpz_mex('inv_logdet',X);
end
