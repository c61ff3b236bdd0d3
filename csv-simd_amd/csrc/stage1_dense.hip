// stage1_dense.hip — the second compilation of stage1_kernels.hip: the stage-1 kernel for delimiter-dense data.
//
// Same source, same algorithm, same tape; another geometry and another emit path:
//   * 2 rounds per wave instead of 8 (64-KiB tiles instead of 256-KiB): a stream that writes more than it reads runs
//     faster the smaller the unit a workgroup draws from the ticket (measured on the 1024 x 4 corpus, round 4:
//     8 / 4 / 2 / 1 rounds = 25.7 / 26.6 / 27.8 / 20.2 % of 8 TB/s read; the 64-column corpus: 65.9 / 57.8 / 43.5 / 25.8 —
//     which is why this is a second instantiation, chosen per launch by the data's density, and not the default);
//   * the DENSE emit path: an 8-KiB window per wave, no per-entry capacity tests (stage1_kernels.hip).
// Only stage1_kernel<true, 0, 0, false, true> and its launcher are compiled here, in namespace csvsimd_dense (the two other
// dense instantiations have files of their own: stage1_dense_d1.hip, stage1_dense_batch.hip).
#define CSVSIMD_DENSE_TU 1
#define CSVSIMD_DENSE_WHICH 0
#define CSVSIMD_ROUNDS 2
#include "stage1_kernels.hip"
