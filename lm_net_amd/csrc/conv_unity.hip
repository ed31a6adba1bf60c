// The dense-convolution family as ONE translation unit (`make timing`: the debug build with the phase clocks, whose device arrays
// must exist once).  The product library builds the five files separately (Makefile).
#include "conv_fwd.hip"
#include "conv_tile_1x1.hip"
#include "conv_tile_3x3.hip"
#include "conv_tileM.hip"
#include "conv_wgrad.hip"
#include "conv_dma3.hip"
#include "conv_dma1.hip"
#include "conv_dmaM.hip"
