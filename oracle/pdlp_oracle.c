/* pdlp_oracle.c -- CPU ORACLE for the restarted-PDHG hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's algorithm (SimplySnap/torchPDLP, the live
 * package under /root/reference/PDLP), function by function, each citing the reference
 * file:line it follows.  It exists to CHECK the HIP path and to serve as bench.py's
 * `cpu_baseline` ("port").  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it; the product (torchpdlp_amd/) never does.
 *
 * Pinning: the reference has no tests or golden vectors of its own, so the
 * oracle is pinned against outputs of the reference itself run in the build container:
 * the .npz files under tests/golden/, produced by tests/golden/gen_golden.py (committed).  tests/test_oracle.py
 * checks every function below against those vectors.
 *
 * Build: see oracle/Makefile  (gcc -O2 -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define REAL float
#define SUF(x) x##_f32
#include "pdlp_oracle_impl.inc"
#undef REAL
#undef SUF

#define REAL double
#define SUF(x) x##_f64
#include "pdlp_oracle_impl.inc"
#undef REAL
#undef SUF

/* number of threads the parallel loops use (1 = the scalar port); returns what is in effect */
int orc_set_threads(int t)
{
#ifdef _OPENMP
    if (t > 0) omp_set_num_threads(t);
    return omp_get_max_threads();
#else
    (void)t;
    return 1;
#endif
}

/* Baseline plumbing (no reference counterpart): copy `n` elements of `elem` bytes with the loops' own static schedule, so that on a
 * NUMA host every page of a matrix array is first touched by the thread that will stream it (dst must be freshly allocated and
 * untouched).  Without it all pages sit on the node of the thread that filled the arrays and a many-core baseline measures one
 * memory controller. */
void orc_spread_copy(void* dst, const void* src, int64_t n, int elem)
{
    const int64_t chunk = 1 << 16;                     /* elements per piece */
    const int64_t pieces = (n + chunk - 1) / chunk;
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < pieces; ++p) {
        const int64_t a = p * chunk, b = (a + chunk < n) ? a + chunk : n;
        const char* s = (const char*)src + a * elem;
        char* d = (char*)dst + a * elem;
        for (int64_t i = 0; i < (b - a) * elem; ++i) d[i] = s[i];
    }
}
