// bfloat16-map build of the affine sampler (see dspn_store.h): the `*_bf16` entry points of include/dspn_nn.h
#define DSPN_HALF 1
#include "sampler.hip"
