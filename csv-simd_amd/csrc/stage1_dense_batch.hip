// stage1_dense_batch.hip — the dense geometry (stage1_dense.hip) for MANY buffers in one launch: stage1_kernel<true, 0, 0, true,
// true> and its launcher only.  A translation unit of its own so that its presence does not change its siblings' code.
#define CSVSIMD_DENSE_TU 1
#define CSVSIMD_DENSE_WHICH 2
#define CSVSIMD_ROUNDS 2
#include "stage1_kernels.hip"
