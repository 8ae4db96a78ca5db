// lr_sched.cc — host-only restatement of torch.optim.lr_scheduler.CosineAnnealingLR in its
// recursive ("chainable") form, which is what every reference agent attaches to its
// optimisers (reference src/agent.py:51-65, :430-444, :825-837, :1203-1212).  The reference
// calls scheduler.step() once per optimiser step, so lr(k+1) is a function of lr(k).
// Double precision, like the Python floats torch keeps in param_groups.
#include "common.h"

#include <cmath>

namespace gcrl {
std::string& last_error() {
  static thread_local std::string msg;
  return msg;
}

int fail(int status, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error() = buf;
  return status;
}
}  // namespace gcrl

extern "C" {

const char* gcrl_last_error(void) { return gcrl::last_error().c_str(); }
int gcrl_abi_version(void) { return GCRL_ABI_VERSION; }

// lr produced by the scheduler.step() call that sets last_epoch = e (e >= 1), given the lr
// currently in the param group.
double gcrl_cosine_lr_next(double lr_now, double base_lr, double eta_min, int64_t t_max,
                           int64_t e) {
  const double pi = 3.141592653589793;  // math.pi
  if (t_max < 1 || e < 1) return lr_now;
  int64_t period = 2 * t_max;
  int64_t phase = (e - 1 - t_max) % period;
  if (phase < 0) phase += period;  // Python's modulo
  if (phase == 0)
    return lr_now + (base_lr - eta_min) * (1.0 - std::cos(pi / (double)t_max)) / 2.0;
  return (1.0 + std::cos(pi * (double)e / (double)t_max)) /
             (1.0 + std::cos(pi * (double)(e - 1) / (double)t_max)) * (lr_now - eta_min) +
         eta_min;
}

}  // extern "C"
