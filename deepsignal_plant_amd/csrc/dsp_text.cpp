// dsp_text.cpp -- host-side text I/O of the call_mods path (feature-TSV parser, per-read-call formatter).
// Filled in below; kept as a separate translation unit because it is plain C++ (no HIP).
#include "dsp_amd.h"
