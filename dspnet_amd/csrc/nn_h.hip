// bfloat16-activation build of the HBM-bound kernels (see dspn_store.h): the `*_bf16` entry points of include/dspn_nn.h
#define DSPN_HALF 1
#include "nn.hip"
