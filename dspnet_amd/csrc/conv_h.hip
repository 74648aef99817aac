// bfloat16-tensor build of the convolution family (see dspn_store.h): the `*_bf16` entry points of include/dspn_nn.h
#define DSPN_HALF 1
#ifndef DSPN_WG_PK
#define DSPN_WG_PK 64      /* weight gradient: 64 pixels per k-step (8 MFMAs per barrier instead of 4) */
#endif
#include "conv.hip"
