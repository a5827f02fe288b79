// Launch plans: a recorded step's launch sequence replayed by one call (include/fs2hip.h, "Launch plans").
// Host code only.  The per-entry-point unpacking lines are generated from the header (plan_thunks.inc).
#include <string.h>

#include "common.h"

static inline float fs2_plan_float(unsigned long long slot) {
  uint32_t b = (uint32_t)slot;
  float f;
  memcpy(&f, &b, 4);
  return f;
}

static const char* const kPlanOpNames[] = {
#define FS2_PLAN_OP(id, name, call) #name,
#include "plan_thunks.inc"
#undef FS2_PLAN_OP
};
static const int kPlanOps = (int)(sizeof(kPlanOpNames) / sizeof(kPlanOpNames[0]));

static inline int fs2_plan_dispatch(int op, const unsigned long long* a, void* s) {
  switch (op) {
#define FS2_PLAN_OP(id, name, call) \
  case id:                          \
    return call;
#include "plan_thunks.inc"
#undef FS2_PLAN_OP
    default:
      return FS2HIP_EINVAL;
  }
}

extern "C" int fs2hip_plan_op_count(void) { return kPlanOps; }

extern "C" int fs2hip_plan_op_id(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < kPlanOps; ++i)
    if (strcmp(name, kPlanOpNames[i]) == 0) return i;
  return -1;
}

extern "C" int fs2hip_plan_events_create(void** out, int n) {
  if (!out || n < 0) return FS2HIP_EINVAL;
  for (int i = 0; i < n; ++i) {
    hipEvent_t e;
    hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (rc != hipSuccess) {
      for (int j = 0; j < i; ++j) (void)hipEventDestroy((hipEvent_t)out[j]);
      return (int)rc;
    }
    out[i] = (void*)e;
  }
  return 0;
}

extern "C" int fs2hip_plan_events_destroy(void* const* events, int n) {
  if (!events || n < 0) return FS2HIP_EINVAL;
  int rc = 0;
  for (int i = 0; i < n; ++i) {
    hipError_t e = hipEventDestroy((hipEvent_t)events[i]);
    if (e != hipSuccess && rc == 0) rc = (int)e;
  }
  return rc;
}

extern "C" int fs2hip_memset(void* dst, int byte, long long nbytes, void* stream) {
  if (nbytes <= 0) return 0;
  if (!dst) return FS2HIP_EINVAL;
  return (int)hipMemsetAsync(dst, byte, (size_t)nbytes, (hipStream_t)stream);
}

extern "C" int fs2hip_plan_replay(const Fs2PlanCmd* cmds, int first, int last, void* const* streams, int n_streams,
                                  void* const* events, int n_events, int* failed_at) {
  if (!cmds || first < 0 || last < first || !streams || n_streams < 1) return FS2HIP_EINVAL;
  const unsigned long long ns = (unsigned long long)n_streams;
  for (int i = first; i < last; ++i) {
    const Fs2PlanCmd& c = cmds[i];
    int rc;
    if (c.op == FS2_PLAN_SYNC) {
      const unsigned long long e = c.a[0], rs = c.a[1], ws = c.a[2];
      if (!events || e >= (unsigned long long)n_events || rs >= ns || ws >= ns) {
        rc = FS2HIP_EINVAL;
      } else {
        rc = (int)hipEventRecord((hipEvent_t)events[e], (hipStream_t)streams[rs]);
        if (rc == 0 && rs != ws) rc = (int)hipStreamWaitEvent((hipStream_t)streams[ws], (hipEvent_t)events[e], 0);
      }
    } else if ((unsigned)c.stream >= (unsigned)n_streams) {
      rc = FS2HIP_EINVAL;
    } else {
      rc = fs2_plan_dispatch(c.op, c.a, streams[c.stream]);
    }
    if (rc != 0) {
      if (failed_at) *failed_at = i;
      return rc;
    }
  }
  return 0;
}
