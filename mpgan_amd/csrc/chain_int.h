// Internal: the specialised schedule of mpg_chain (chain2.hip), tried first by the entry point in chain.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/mpgan_amd.h"
#define MPG_CHAIN2_NA (-100)   // "not one of my shapes": the caller runs the general kernel
int mpg_chain2_try(const MpgChain* p, hipStream_t st);
