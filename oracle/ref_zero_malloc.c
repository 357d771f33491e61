/* oracle/ref_zero_malloc.c — linked ONLY into oracle/_ref/*.so via -Wl,--wrap=malloc (test infrastructure).
 * With CPU_GEMM=1 the reference scales the UNINITIALISED gemm output by beta = 0 before accumulating
 * (/root/reference/Executable/gemm.c:66-67), so heap garbage that happens to be Inf/NaN (e.g. 0xFFFFFFFF words)
 * poisons the result: the same inputs gave run-to-run different masks.  The author's MKL build never reads C when
 * beta = 0.  Handing the reference zeroed buffers pins the MKL semantics without touching its sources. */
#include <stdlib.h>
void *__wrap_malloc(size_t n) { return calloc(1, n); }
