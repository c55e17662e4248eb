// The pendulum instantiations of the reverse sweep as their own translation unit (see the note in rollout_bwd.hip): same source,
// the scheduling flags that suit THEM (mpg_amd/build.py).
#define MPG_BWD_PENDULUM_PART 1
#include "rollout_bwd.hip"
