// The pendulum instantiations of the forward sweep as their own translation unit (see the note in rollout_fwd.hip).
#define MPG_FWD_PENDULUM_PART 1
#include "rollout_fwd.hip"
