// Entry points declared in include/metafem_mi355x.h whose kernels are not written yet: they fail
// loudly (MFEM_ERR_UNSUPPORTED + message) instead of falling back to anything.
#include "krylov.h"

#define UNSUP(name)                                   \
  do {                                                \
    mfem_set_error(name " is not implemented yet");   \
    return MFEM_ERR_UNSUPPORTED;                      \
  } while (0)

extern "C" int mfem_op_var(mfem_context, const mfem_op_layout*, const double*, int32_t, int64_t, const int32_t*, const double*, double*, const int32_t*, const int32_t*, int64_t) { UNSUP("mfem_op_var"); }
extern "C" int mfem_op_kval(mfem_context, const mfem_op_layout*, const double*, int32_t, int32_t, const double*, const int32_t*, int64_t, double*, const int32_t*, const int32_t*, int64_t) { UNSUP("mfem_op_kval"); }
extern "C" int mfem_op_res(mfem_context, const mfem_op_layout*, const double*, int32_t, const double*, int64_t, const int32_t*, double*, const int32_t*, const int32_t*, int64_t) { UNSUP("mfem_op_res"); }
