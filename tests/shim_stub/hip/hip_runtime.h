/* TEST INFRASTRUCTURE ONLY -- a stand-in for <hip/hip_runtime.h> with the handful of names csrc/fpga_shim.cpp uses, so
 * that the shim's host code (threads, queues, pinned-buffer pool, packet assembly) can be built with g++ and run under
 * ThreadSanitizer / AddressSanitizer in the CPU container (GPU sanitizers are not available on the pool).  "Pinned" memory
 * is plain malloc here.  Nothing in the product includes this file. */
#ifndef CHAINDP_TEST_HIP_STUB_H
#define CHAINDP_TEST_HIP_STUB_H
#include <stdlib.h>
typedef int hipError_t;
#define hipSuccess 0
#define hipHostMallocDefault 0
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned flags) { (void)flags; *p = malloc(n ? n : 1); return *p ? 0 : 2; }
static inline hipError_t hipHostFree(void *p) { free(p); return 0; }
#endif
