/* device_launch_parameters.h shim (see cuda_runtime.h) */
#pragma once
#include "cuda_runtime.h"
