// Host/device qualifier shared by the small algorithm headers that are compiled both by hipcc
// (device code of liborbhip.so) and by g++ (host unit tests of the same functions).
#ifndef ORBHIP_HD_H
#define ORBHIP_HD_H
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ORB_HD __host__ __device__
#else
#define ORB_HD
#endif
#endif
