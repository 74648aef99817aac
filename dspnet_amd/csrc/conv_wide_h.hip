// the bfloat16-tensor build of conv_wide.hip (csrc/dspn_store.h): the wide tile family on bf16 activations and weight copies
#define DSPN_HALF
#include "conv_wide.hip"
