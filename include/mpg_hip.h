/*
 * mpg_hip.h - C ABI of libmpg_hip.so, the MI355X (gfx950) drop-in for the data-parallel hot path of
 * idthanm/mpg (Mixed Policy Gradient).
 *
 * The reference has no FFI: its "plugin API" is a set of duck-typed Python classes picked through
 * registries (train_scripts/train_script.py:39-51, envs_and_models/__init__.py:13-15).  Each entry point
 * below replaces the arithmetic of one group of those methods; the file:line it replaces is cited on
 * every declaration (paths relative to the reference root).  The python package mpg_amd keeps the reference's class and
 * method names on top of this ABI; INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. a torch tensor), contiguous float32
 *     unless stated; nothing is allocated, freed or copied to the host behind the caller's back;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is enqueued on it and
 *     the call returns immediately (no host synchronisation);
 *   - return value: 0 on success, a negative MPG_E* code or -(hipError_t) otherwise; no exceptions cross
 *     the boundary;
 *   - re-entrant per (device, stream): no global mutable state;
 *   - networks are 2-hidden-layer ELU MLPs with 256 hidden units (model.py:20-43), parameters stored as
 *     ONE flat float32 vector in Keras order  W1[in][256] b1[256] W2[256][256] b2[256] W3[256][out] b3[out]
 *     (kernels row-major (in,out), exactly `Model.get_weights()` flattened);
 *   - row batches (obs, act, ...) are row-major [rows][dim] like the reference's numpy arrays.
 */
#ifndef MPG_HIP_H
#define MPG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mpg_stream_t; /* hipStream_t */

enum {
    MPG_OK = 0,
    MPG_EINVAL = -1000,   /* bad argument (null pointer, unsupported dimension, ...) */
    MPG_EWORKSPACE = -1001 /* workspace too small: call the *_workspace_bytes query */
};

enum { MPG_ENV_PATH_TRACKING = 0, MPG_ENV_INVERTED_PENDULUM = 1 };
enum { MPG_ACT_LINEAR = 0, MPG_ACT_TANH = 1 };
enum { MPG_HIDDEN = 256 };
/* floats per agent of the opaque env state block, stored SoA [MPG_ENV_STATE_DIM][n]:
 * v_x, v_y, r, y, phi, x (veh_full_state) + delta_y, delta_phi (the two derived entries of veh_state). */
enum { MPG_ENV_STATE_DIM = 8 };

/* ABI version of this header; bumped on any signature change. */
int mpg_abi_version(void);
/* Human-readable description of the last error on this thread ("" if none). */
const char* mpg_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Vectorised real environment (K1)
 * ---------------------------------------------------------------------------------------------- */

/* PathTrackingEnv.reset(init_obs=...)  - envs_and_models/path_tracking_env.py:410-421.
 * Rebuilds the full state of all n agents from obs [n][6] (= [v_x-20, v_y, r, dy, dphi, x]). */
int mpg_env_reset_from_obs(int env_kind, int n, float* state, const float* init_obs, mpg_stream_t stream);

/* PathTrackingEnv.reset()  - path_tracking_env.py:423-454.  Re-draws every agent whose done_mask byte is
 * non-zero (done_mask == NULL: all agents) from the reset law x~U(0,600), dy~N(0,1), dphi~N(0,pi/9),
 * v_x~U(15,25), beta~N(0,.15), v_y=v_x tan(beta), r~N(0,.3) with a counter-based Philox4x32-10 stream
 * keyed by (seed, call counter ctr, agent index); writes obs [n][6] for ALL agents. */
int mpg_env_reset(int env_kind, int n, float* state, const uint8_t* done_mask, uint64_t seed, uint64_t ctr,
                  float* obs, mpg_stream_t stream);

/* PathTrackingEnv.step  - path_tracking_env.py:456-487 with VehicleDynamics.simulation :144-179 (20
 * sub-steps at 200 Hz), compute_rewards :181-199 (on the pre-step state), judge_done :474-487.
 * action [n][2] in [-1,1] (scaled by [1.2pi/9, 3] and clipped inside, :457-459); outputs obs [n][6],
 * reward [n], done [n] bytes.  done follows the reference literally and is therefore always 1
 * (SURVEY.md B-0); done_intended (nullable) receives the evidently intended |alpha| > |bound| test. */
int mpg_env_step(int env_kind, int n, float* state, const float* action, float* obs, float* reward,
                 uint8_t* done, uint8_t* done_intended, mpg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MPG_HIP_H */
