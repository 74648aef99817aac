// PRIVATE to dspnet_amd/csrc (not part of the C ABI in include/): timing-only ablation switches of conv_nt_kernel.
// They exist only in a library built with `make ABLATE=1` (-DDSPN_ABLATE -> ../libdspn_hip_ablate.so), which the
// experiment scripts under scratch/ load with DSPN_LIB=...; the production library has neither the symbol nor the
// branches.  Results are WRONG when any bit is set.
//   bit 1 (2): skip the LDS stores   bit 2 (4): skip the per-k-step barrier   bit 4 (16): skip the output stores
//   bit 5 (32): skip the epilogue    bits 8-10: force a tile configuration    bit 11 (2048): in-kernel clock stamps
//   bit 14 (16384): per-wave phase stamps of the k-step (conv_nt_kernel, split modes)
//   bit 15 (32768): per-wave phase stamps of the tile EPILOGUE (conv_nt_kernel): staging | barrier | row loop | statistics
//                   exchange | hand-over to the next tile | the tile's k-loop; read like bit 14's (scratch/epi_stamps.py)
//   bit 12 (4096) / bit 13 (8192): the A side of a split-mode k-step (loads, affine, pieces, LDS stores) only every 3rd / 9th k-step
#pragma once
extern "C" int dspn_debug_set(int bits);
extern "C" int dspn_debug_read_stamps(unsigned *host, int words, int clear);
