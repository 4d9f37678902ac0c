// The generic implicit-GEMM convolution of conv_igemm.hip as a GROUPED convolution: one grid row per group, the group
// strides as extra kernel arguments (see the macro block at the top of conv_igemm.hip).  replaces nn.Conv2d(groups = g), the
// 3x3 of a RegNet bottleneck (empanada/models/encoders/regnet.py:51-77 via blocks.py:134-153), on the fp16 engine: one
// launch instead of one per group (203 / 180 launches per forward of regnetx_6p4gf / regnety_6p4gf).
#define EMP_IGEMM_GROUPED 1
#include "conv_igemm.hip"
