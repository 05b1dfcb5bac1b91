// wn_gemm.h -- argument block of the generic fp32-MFMA GEMM (wn_gemm.hip).
#pragma once
#include "wn_device.h"

// The argument block is part of the public C ABI (wn_op_gemm): include/wavenet_hip_gemm.h.
#include "../../include/wavenet_hip_gemm.h"

// returns 0 on success
int wn_gemm_launch(const WnGemmArgs* g, wn_stream_t stream);
