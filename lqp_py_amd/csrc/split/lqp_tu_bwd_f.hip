// the backward's Cholesky kernel with its tile products on the float16 matrix pipe (two-half operands, lqp_f16x2.hpp)
#include "../lqp_unroll.hpp"
template __global__ void lqp::k_bwd_chol_solve<0, true>(lqp::BwdParams<float>);
template __global__ void lqp::k_bwd_chol_solve<4, true>(lqp::BwdParams<float>);
