/* The GPU box's own glibc, callable over arrays (test infrastructure for tests/test_libm.py::test_device_libm_matches_host_glibc).
 *   gcc -O1 -fno-builtin -shared -fPIC host_libm.c -o host_libm.so -lm */
#define _GNU_SOURCE
#include <math.h>
void host_libm_eval(int fn, const float *a, const float *b, long n, float *out, float *out2) {
    for (long i = 0; i < n; ++i) {
        float x = a[i], r = 0, r2 = 0;
        switch (fn) {
        case 0: r = sinf(x); break;
        case 1: r = cosf(x); break;
        case 2: sincosf(x, &r, &r2); break;
        case 3: r = expf(x); break;
        case 4: r = logf(x); break;
        case 5: r = acosf(x); break;
        case 6: r = atanf(x); break;
        case 7: r = atan2f(x, b[i]); break;
        }
        out[i] = r;
        if (out2) out2[i] = r2;
    }
}
