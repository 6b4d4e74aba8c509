#pragma once
#include "../../include/mpgan_amd.h"
