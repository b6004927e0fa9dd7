// IMC cost / gradient / Hessian-vector kernels (placeholder translation unit; filled in below).
#include "alsmi_common.h"
