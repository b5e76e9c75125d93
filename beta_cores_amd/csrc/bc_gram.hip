#include "bc_internal.h"
extern "C" int bc_weighted_gram(bc_ctx* ctx, const bc_data* data, const double* w, double* out_xtwx, double* out_xtwy){ bc_set_error("bc_weighted_gram: not built yet"); return -1; }
