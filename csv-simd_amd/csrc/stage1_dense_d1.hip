// stage1_dense_d1.hip — the dense geometry (stage1_dense.hip) for another delimiter / quote byte: stage1_kernel<true, 0, 1,
// false, true> and its launcher only.  A translation unit of its own so that its presence does not change its siblings' code.
#define CSVSIMD_DENSE_TU 1
#define CSVSIMD_DENSE_WHICH 1
#define CSVSIMD_ROUNDS 2
#include "stage1_kernels.hip"
