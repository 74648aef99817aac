// bfloat16-tensor build of the convolution family (see dspn_store.h): the `*_bf16` entry points of include/dspn_nn.h
// (the weight gradient keeps 32 pixels per k-step: 64 needs 80 KB of LDS per workgroup, one workgroup per CU, and measured
// 3.79 ms against 2.85 ms per resnet-50 step)
#define DSPN_HALF 1
#include "conv.hip"
