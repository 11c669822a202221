// Library instantiation of the FP64 MFMA GEMM kernels (gemm_f64_kernel.hpp).
#include "gemm_f64_kernel.hpp"
