// bfloat16-tensor build of the convolution family (see dspn_store.h): the `*_bf16` entry points of include/dspn_nn.h
#define DSPN_HALF 1
#include "conv.hip"
