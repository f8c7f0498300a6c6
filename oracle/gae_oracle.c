/*
 * oracle/gae_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's GAE backward recurrence
 * (rlgym_ppo/util/torch_functions.py:36-78, called from rlgym_ppo/learner.py:358-366).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product path (rlgym_ppo_amd) never does.
 *
 * The reference runs the recurrence on Python/NumPy scalars, so its intermediate precision is an
 * accident of NumPy's scalar promotion rules.  Two modes restate the two
 * behaviours that exist in the wild:
 *
 *   mode 0  "np1/f64": NumPy < 2.0 (the reference's pin, requirements.txt:7).  Value-based scalar
 *           promotion makes every product with a Python float a float64, so the whole recurrence
 *           runs in double; only `rews[step] / return_std` (float32 / float32,
 *           torch_functions.py:63) and `1 - dones[step]` stay float32.
 *   mode 1  "np2": NumPy >= 2.0 (NEP 50 weak scalars; what runs in the build container and what the
 *           golden vectors tests/golden/g3_gae.npz were produced with).  A Python float meeting a
 *           float32 scalar is first rounded to float32, so `delta` is float32 arithmetic while
 *           `ret` / `adv` accumulate in double *iff* `truncated` arrived as a float64 array
 *           (batched_agent_manager.py:145 makes it one); with a float32 `truncated` everything that
 *           touches a float32 scalar stays float32.
 *
 * Pinning: mode 1 is checked bit-for-bit against g3_gae.npz (reference run here, NumPy 2.2.6);
 * mode 0 is what the HIP kernel implements and is checked against the same vectors to
 * atol 1e-5 + rtol 1e-5 (SURVEY.md section 8(c) "Tolerances").
 *
 * Outputs: value_targets and advantages are float32 (torch.as_tensor(..., dtype=float32),
 * torch_functions.py:76-77); returns are the un-rounded scalars (a Python list in the reference).
 */
#include <math.h>
#include <stddef.h>

static float clipf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* trunc64 != NULL : truncated given as float64 (reference default); else trunc32 is used. */
int gae_oracle(const float *rews, const float *dones, const double *trunc64, const float *trunc32,
               const float *values /* N+1 */, long n, double gamma, double lmbda,
               int use_std, float return_std, int mode,
               float *value_targets, float *advantages, double *returns)
{
    if (n < 0) return 1;
    if (mode == 0) {
        double last_adv = 0.0, last_ret = 0.0;
        for (long t = n - 1; t >= 0; --t) {
            float nd32 = 1.0f - dones[t];                       /* torch_functions.py:59 */
            double nd = (double)nd32;
            double nt = trunc64 ? 1.0 - trunc64[t] : (double)(1.0f - trunc32[t]); /* :60 */
            double r = (double)rews[t];
            double rn = use_std ? (double)clipf(rews[t] / return_std, -10.0f, 10.0f) : r; /* :62-65 */
            double pred = rn + gamma * (double)values[t + 1] * nd;                /* :67 */
            double delta = pred - (double)values[t];                              /* :68 */
            double ret = r + last_ret * gamma * nd * nt;                          /* :69 */
            returns[t] = ret;
            last_ret = ret;
            last_adv = delta + gamma * lmbda * nd * nt * last_adv;                /* :72 */
            advantages[t] = (float)last_adv;                                      /* :76 */
            value_targets[t] = (float)((double)values[t] + last_adv);             /* :77 */
        }
        return 0;
    }
    if (mode == 1 && trunc64) {
        /* NEP-50 promotions with a float64 `truncated` array. */
        double last_adv = 0.0, last_ret = 0.0;
        float gl32 = (float)(gamma * lmbda);   /* python float product, rounded when it meets not_done */
        for (long t = n - 1; t >= 0; --t) {
            float nd = 1.0f - dones[t];
            double nt = 1.0 - trunc64[t];
            float rn = use_std ? clipf(rews[t] / return_std, -10.0f, 10.0f) : rews[t];
            float gv = (float)(gamma * (double)values[t + 1]);  /* py float * py float, then weak -> f32 */
            float pred = rn + gv * nd;
            float delta = pred - values[t];
            /* last_return * gamma is float64 (np.float64 * py float) from the 2nd iteration on; on the
             * first it is the python float 0.0 -- same value. (x * nd) promotes nd to float64. */
            double ret = (double)rews[t] + ((last_ret * gamma) * (double)nd) * nt;
            returns[t] = ret;
            last_ret = ret;
            double coef = (double)(gl32 * nd) * nt;             /* (py*f32 -> f32) * f64 -> f64 */
            last_adv = (double)delta + coef * last_adv;
            advantages[t] = (float)last_adv;
            value_targets[t] = (float)((double)values[t] + last_adv);
        }
        return 0;
    }
    if (mode == 1) {
        /* NEP-50 promotions with a float32 `truncated` array: python ints/floats are weak everywhere,
         * so every intermediate is float32 -- except the very first iteration's `0 * gamma`, a python
         * float that is rounded to float32 on contact (value 0 either way). */
        float last_adv = 0.0f, last_ret = 0.0f;
        float gl32 = (float)(gamma * lmbda);
        float g32 = (float)gamma;
        for (long t = n - 1; t >= 0; --t) {
            float nd = 1.0f - dones[t];
            float nt = 1.0f - trunc32[t];
            float rn = use_std ? clipf(rews[t] / return_std, -10.0f, 10.0f) : rews[t];
            float gv = (float)(gamma * (double)values[t + 1]);
            float pred = rn + gv * nd;
            float delta = pred - values[t];
            float ret = rews[t] + ((last_ret * g32) * nd) * nt;
            returns[t] = (double)ret;
            last_ret = ret;
            last_adv = delta + ((gl32 * nd) * nt) * last_adv;
            advantages[t] = last_adv;
            value_targets[t] = values[t] + last_adv;                /* py float (weak) + np.float32 -> f32 add */
        }
        return 0;
    }
    return 2;
}
