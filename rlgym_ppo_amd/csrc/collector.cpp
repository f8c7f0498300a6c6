// collector.cpp -- [r6] the learner-side loop of the reference's process-per-environment collection in C++ (host code, no GPU):
// what rlgym_ppo/batched_agents/batched_agent_manager.py:126-350 + batched_trajectory.py:58-105 do per worker message in Python --
// wait for the ready sockets, read the datagram, parse the worker's slab of the shared array, advance the observation statistics,
// standardise, keep the episode-reward bookkeeping, bank the timestep into the worker's trajectory, and at the end of a
// collect_timesteps call lay all trajectories out agent by agent -- behind the same wire format (comm_consts.py) and with the same
// observable results, value for value (tests/test_wire_format.py, tests/test_native_collector.py).  The policy call stays in Python:
// per inference the host makes three calls here (ready -> [get_action] -> send -> collect) instead of ~20 interpreter-level
// operations per worker message.  50,000 timesteps of 8 two-agent workers: 27,000 messages, 0.32 s of the collection's 0.70 s.
#ifndef _GNU_SOURCE
#define _GNU_SOURCE 1
#endif
#include <errno.h>
#include <netinet/in.h>
#include <poll.h>
#include <string.h>
#include <sys/socket.h>

#include <cmath>
#include <new>
#include <vector>

#include "../../include/rlppo.h"

namespace rlppo {
void set_error(const char *fmt, ...);  // api.hip: the text behind rlppo_last_error()
}

#define COLLECTOR_FAIL(code, ...)      \
    do {                               \
        rlppo::set_error(__VA_ARGS__); \
        return code;                   \
    } while (0)

namespace {
constexpr float STEP_HEADER0 = 83775.f;                                 // comm_consts.ENV_STEP_DATA_HEADER[0]
constexpr float ACTIONS_HEADER[3] = {12782.f, 83783.f, 80784.f};        // comm_consts.POLICY_ACTIONS_HEADER
constexpr int PACKET_MAX = 8192;                                        // comm_consts.PACKET_MAX_SIZE

struct Step {  // one banked timestep of one environment (BatchedTrajectory.complete_timesteps entry)
    int n_state = 0, n_next = 0;
    std::vector<float> state, action, logp, next;
    std::vector<double> rew;
    double done = 0, trunc = 0;
};
struct Worker {
    int fd = -1;
    sockaddr_in peer{};
    const float *slab = nullptr;
    int cur_n = -1, next_n = -1;  // rows of current_obs / next_obs; -1 = None
    std::vector<float> cur_obs, next_obs;
    // the pending timestep's fields (BatchedTrajectory.state ... .truncated); has_* = "is not None"
    bool has_state = false, has_reward = false, has_next = false, has_done = false;
    int p_n = 0, p_next_n = 0;
    std::vector<float> p_state, p_action, p_logp, p_next;
    std::vector<double> p_rew;
    double p_done = 0, p_trunc = 0;
    std::vector<Step> traj;
    std::vector<double> ep{0.0};
};
struct Metrics {
    std::vector<float> values;
    std::vector<int> shape;
};
struct Collector {
    int n = 0, d = 0, act_width = 0;
    int64_t slab_floats = 0;
    std::vector<Worker> w;
    std::vector<int> current_pids, ready_pids;
    std::vector<std::vector<Step>> completed;
    std::vector<Metrics> metrics;
    bool avg_none = true;
    double avg = 0.0;
};

// BatchedTrajectory.update(): bank the pending timestep if all of its fields are there; true when it ended the episode
bool traj_update(Worker &w) {
    if (!(w.has_state && w.has_reward && w.has_next && w.has_done)) return false;
    Step s;
    s.n_state = w.p_n;
    s.n_next = w.p_next_n;
    s.state.swap(w.p_state);
    s.action.swap(w.p_action);
    s.logp.swap(w.p_logp);
    s.next.swap(w.p_next);
    s.rew.swap(w.p_rew);
    s.done = w.p_done;
    s.trunc = w.p_trunc;
    w.traj.push_back(std::move(s));
    w.has_state = w.has_reward = w.has_next = w.has_done = false;  // (`truncated` keeps its value, like the reference)
    return w.traj.back().done != 0.0;
}

// WelfordRunningStat.update over the rows of `x`, in the reference's operation order and in the state's dtype (float32 state: every
// step rounds where numpy's float32 arithmetic rounds; float64 state -- statistics restored from JSON -- likewise)
template <typename T>
void welford_rows(const float *x, int rows, int d, T *mean, T *var, int64_t *count) {
    for (int r = 0; r < rows; ++r) {
        const int64_t prev = *count;
        *count = prev + 1;
        const T cn = (T)*count, cp = (T)prev;
        for (int k = 0; k < d; ++k) {
            const T delta = (T)x[(size_t)r * d + k] - mean[k];
            const T delta_n = delta / cn;
            mean[k] += delta_n;
            var[k] += delta * delta_n * cp;
        }
    }
}
}  // namespace

extern "C" {

int rlppo_collector_create(int32_t n_workers, const int32_t *socket_fds, const int32_t *peer_ports, const float *shm_base,
                           int64_t shm_floats_per_worker, int32_t obs_dim, void **handle) {
    if (!handle || n_workers <= 0 || !socket_fds || !peer_ports || !shm_base || shm_floats_per_worker < 8 || obs_dim <= 0)
        COLLECTOR_FAIL(RLPPO_ERR_ARG, "collector_create: bad argument (n_workers=%d, obs_dim=%d, %ld floats per slab)", n_workers, obs_dim, (long)shm_floats_per_worker);
    Collector *c = new (std::nothrow) Collector();
    if (!c) COLLECTOR_FAIL(RLPPO_ERR_ARG, "collector_create: out of memory");
    c->n = n_workers;
    c->d = obs_dim;
    c->slab_floats = shm_floats_per_worker;
    c->w.resize(n_workers);
    for (int i = 0; i < n_workers; ++i) {
        Worker &w = c->w[i];
        w.fd = socket_fds[i];
        w.peer.sin_family = AF_INET;
        w.peer.sin_port = htons((uint16_t)peer_ports[i]);
        w.peer.sin_addr.s_addr = htonl(INADDR_LOOPBACK);  // workers live on 127.0.0.1 (batched_agent_manager.py:436-470)
        w.slab = shm_base + (size_t)i * shm_floats_per_worker;
    }
    *handle = c;
    return 0;
}

int rlppo_collector_destroy(void *handle) {
    delete static_cast<Collector *>(handle);
    return 0;
}

// current_obs[worker] <- obs (the worker's reset state, as the handshake received it: NOT standardised, like the reference's);
// ready != 0 also appends the worker to current_pids
int rlppo_collector_set_obs(void *handle, int32_t worker, const float *obs, int32_t rows, int32_t ready) {
    Collector *c = static_cast<Collector *>(handle);
    if (!c || worker < 0 || worker >= c->n || rows < 0 || (rows > 0 && !obs)) COLLECTOR_FAIL(RLPPO_ERR_ARG, "collector_set_obs: bad argument (worker %d, %d rows)", worker, rows);
    Worker &w = c->w[worker];
    w.cur_obs.assign(obs, obs + (size_t)rows * c->d);
    w.cur_n = rows;
    if (ready) {
        bool in = false;
        for (int p : c->current_pids) in |= p == worker;
        if (!in) c->current_pids.push_back(worker);
    }
    return 0;
}

// _send_actions, first half: the stacked observations of the workers that wait for actions (current_pids that have an observation)
int rlppo_collector_ready(void *handle, float *obs_out, int64_t cap_rows, int64_t *n_rows) {
    Collector *c = static_cast<Collector *>(handle);
    if (!c || !n_rows) COLLECTOR_FAIL(RLPPO_ERR_ARG, "collector_ready: null argument");
    c->ready_pids.clear();
    int64_t rows = 0;
    for (int pid : c->current_pids)
        if (c->w[pid].cur_n >= 0) {
            if (rows + c->w[pid].cur_n > cap_rows) COLLECTOR_FAIL(RLPPO_ERR_WORKSPACE, "collector_ready: more than %ld waiting observations", (long)cap_rows);
            if (c->w[pid].cur_n > 0) memcpy(obs_out + rows * c->d, c->w[pid].cur_obs.data(), (size_t)c->w[pid].cur_n * c->d * sizeof(float));
            rows += c->w[pid].cur_n;
            c->ready_pids.push_back(pid);
        }
    *n_rows = rows;
    return 0;
}

// _send_actions, second half: row r of the batch handed out by the last _ready call gets actions[r][act_width] / log_probs[r]; the
// pending timestep of each ready worker records (state, action, log_prob) and the worker is sent POLICY_ACTIONS_HEADER + its rows
int rlppo_collector_send(void *handle, const float *actions, int32_t act_width, const float *log_probs) {
    Collector *c = static_cast<Collector *>(handle);
    if (!c || act_width <= 0 || !actions || !log_probs) COLLECTOR_FAIL(RLPPO_ERR_ARG, "collector_send: bad argument (act_width %d)", act_width);
    if (c->ready_pids.empty()) return 0;
    c->act_width = act_width;
    int64_t row = 0;
    // ONE sendmmsg for all ready workers: a worker answers to the endpoint it was spawned with whatever the source of its actions
    // (batched_agent.py:154-167), so every datagram leaves through the first ready worker's socket -- one system call instead of one
    // per worker (8 workers: 20 us -> 8 us per inference)
    const size_t nr = c->ready_pids.size();
    std::vector<std::vector<float>> msgs(nr);
    std::vector<mmsghdr> hdrs(nr);
    std::vector<iovec> iov(nr);
    for (size_t k = 0; k < nr; ++k) {
        Worker &w = c->w[c->ready_pids[k]];
        const int n = w.cur_n;
        w.p_state = w.cur_obs;  // (the inference batch's rows ARE the current observations)
        w.p_n = n;
        w.p_action.assign(actions + row * act_width, actions + (row + n) * act_width);
        w.p_logp.assign(log_probs + row, log_probs + row + n);
        w.has_state = true;
        msgs[k].assign(ACTIONS_HEADER, ACTIONS_HEADER + 3);
        msgs[k].insert(msgs[k].end(), w.p_action.begin(), w.p_action.end());
        iov[k] = iovec{msgs[k].data(), msgs[k].size() * sizeof(float)};
        memset(&hdrs[k], 0, sizeof(mmsghdr));
        hdrs[k].msg_hdr.msg_name = &w.peer;
        hdrs[k].msg_hdr.msg_namelen = sizeof(w.peer);
        hdrs[k].msg_hdr.msg_iov = &iov[k];
        hdrs[k].msg_hdr.msg_iovlen = 1;
        row += n;
    }
    const int fd = c->w[c->ready_pids[0]].fd;
    for (size_t sent = 0; sent < nr;) {
        const int rc = sendmmsg(fd, hdrs.data() + sent, (unsigned)(nr - sent), 0);
        if (rc < 0) {
            if (errno == EINTR) continue;
            COLLECTOR_FAIL(RLPPO_ERR_ARG, "collector_send: sendmmsg failed: %s", strerror(errno));
        }
        sent += (size_t)rc;
    }
    c->current_pids.clear();
    c->ready_pids.clear();
    return 0;
}

// _collect_responses(min_obs) + the promotion of next_obs + _sync_trajectories of one collect_timesteps iteration.
// resume != 0: the continuation of a wait that returned RLPPO_ERR_INTERRUPTED (min_obs = what is still missing).
// standardize: 0 = off; 1 = (x - mean[0]) / std[0] (the reference's scalars, quirk Q5); 2 = per feature.  stats_*: the
// WelfordRunningStat's arrays (float32 when stats_f64 == 0) and count, advanced in place every steps_per_increment-th message.
int rlppo_collector_collect(void *handle, int64_t min_obs, int32_t resume, int32_t standardize, const void *mean, const void *stdv, void *stats_mean,
                            void *stats_var, int64_t *stats_count, int32_t stats_f64, int64_t steps_per_increment, int64_t *steps_since_increment,
                            int64_t *n_collected) {
    Collector *c = static_cast<Collector *>(handle);
    if (!c || !n_collected || (standardize && (!mean || !stdv || !stats_mean || !stats_var || !stats_count || !steps_since_increment)))
        COLLECTOR_FAIL(RLPPO_ERR_ARG, "collector_collect: null argument");
    if (!resume) c->current_pids.clear();
    std::vector<pollfd> fds(c->n);
    for (int i = 0; i < c->n; ++i) fds[i] = pollfd{c->w[i].fd, POLLIN, 0};
    alignas(16) unsigned char buf[PACKET_MAX];
    int64_t got = 0;
    const int d = c->d;
    while (got < min_obs) {
        int rc = poll(fds.data(), (nfds_t)c->n, 60000);
        if (rc < 0 && errno == EINTR) {  // a signal: hand control back so that the host's handlers run (Ctrl-C, alarms); resume != 0 continues this wait
            *n_collected = got;
            return RLPPO_ERR_INTERRUPTED;
        }
        if (rc < 0) COLLECTOR_FAIL(RLPPO_ERR_ARG, "collector_collect: poll failed: %s", strerror(errno));
        if (rc == 0) COLLECTOR_FAIL(RLPPO_ERR_COLLECT_TIMEOUT, "collector_collect: no worker message for a minute (%ld of %ld agent-steps of this wait arrived)", (long)got, (long)min_obs);
        for (int pid = 0; pid < c->n; ++pid) {
            if ((fds[pid].revents & (POLLERR | POLLHUP | POLLNVAL)) && !(fds[pid].revents & POLLIN))  // a dead socket would spin this loop
                COLLECTOR_FAIL(RLPPO_ERR_ARG, "collector_collect: the socket of worker %d is dead (revents 0x%x)", pid, (unsigned)fds[pid].revents);
            if (!(fds[pid].revents & POLLIN)) continue;
            Worker &w = c->w[pid];
            const ssize_t len = recv(w.fd, buf, sizeof(buf), 0);
            if (len < 12) continue;
            float h0;
            memcpy(&h0, buf, 4);
            if (h0 != STEP_HEADER0) continue;  // (not step data: ignored, like the Python loop)
            // ---- the slab (comm_consts.py): [prev_n, done, truncated, rank(state), rank(metrics), *metrics_shape, *state_shape, *rewards, *metrics, *obs]
            const float *s = w.slab;
            const int prev_n = (int)s[0], state_rank = (int)s[3], metrics_rank = (int)s[4];
            const double done = s[1], trunc = s[2];
            if (prev_n < 0 || state_rank < 1 || state_rank > 2 || metrics_rank < 0 || metrics_rank > 8 || 5 + metrics_rank + state_rank + prev_n > c->slab_floats)
                COLLECTOR_FAIL(RLPPO_ERR_ARG, "collector_collect: worker %d's slab has no valid step header (prev_n %d, state rank %d, metrics rank %d)", pid, prev_n, state_rank, metrics_rank);
            int64_t o = 5;
            Metrics m;
            int64_t n_metrics = metrics_rank ? 1 : 0;
            for (int k = 0; k < metrics_rank; ++k) {
                m.shape.push_back((int)s[o + k]);
                n_metrics *= (int)s[o + k];
            }
            o += metrics_rank;
            const int rows = state_rank == 1 ? 1 : (int)s[o], width = state_rank == 1 ? (int)s[o] : (int)s[o + 1];
            o += state_rank;
            if (width != d || rows < 0 || n_metrics < 0 || o + prev_n + n_metrics + (int64_t)rows * d > c->slab_floats)
                COLLECTOR_FAIL(RLPPO_ERR_ARG, "collector_collect: worker %d's step does not fit (observation %d x %d against width %d, %ld metrics, %ld floats per slab)", pid, rows, width, d, (long)n_metrics, (long)c->slab_floats);
            std::vector<double> rews(s + o, s + o + prev_n);
            o += prev_n;
            m.values.assign(s + o, s + o + n_metrics);
            o += n_metrics;
            c->metrics.push_back(std::move(m));
            std::vector<float> nxt(s + o, s + o + (size_t)rows * d);
            if (standardize) {
                if (*steps_since_increment > steps_per_increment) {
                    if (stats_f64) welford_rows<double>(nxt.data(), rows, d, static_cast<double *>(stats_mean), static_cast<double *>(stats_var), stats_count);
                    else welford_rows<float>(nxt.data(), rows, d, static_cast<float *>(stats_mean), static_cast<float *>(stats_var), stats_count);
                    *steps_since_increment = 0;
                } else {
                    ++*steps_since_increment;
                }
                if (stats_f64) {
                    // statistics restored from JSON are float64 (running_stats.py:120-125): numpy then forms (x - mean) / std in
                    // float64 and the observation is rounded to float32 once, when it meets the policy / the experience buffer
                    const double *m64 = static_cast<const double *>(mean), *s64 = static_cast<const double *>(stdv);
                    for (int r = 0; r < rows; ++r)
                        for (int k = 0; k < d; ++k) {
                            const double mu = standardize == 2 ? m64[k] : m64[0], sd = standardize == 2 ? s64[k] : s64[0];
                            double v = ((double)nxt[(size_t)r * d + k] - mu) / sd;
                            v = v < -5.0 ? -5.0 : (v > 5.0 ? 5.0 : v);
                            nxt[(size_t)r * d + k] = (float)v;
                        }
                } else {
                    const float *m32 = static_cast<const float *>(mean), *s32 = static_cast<const float *>(stdv);
                    for (int r = 0; r < rows; ++r)
                        for (int k = 0; k < d; ++k) {
                            const float mu = standardize == 2 ? m32[k] : m32[0], sd = standardize == 2 ? s32[k] : s32[0];
                            float v = (nxt[(size_t)r * d + k] - mu) / sd;
                            v = v < -5.f ? -5.f : (v > 5.f ? 5.f : v);  // np.clip: NaN stays NaN
                            nxt[(size_t)r * d + k] = v;
                        }
                }
            }
            // episode-reward bookkeeping (python floats: doubles)
            for (int i = 0; i < prev_n; ++i) {
                if (i >= (int)w.ep.size()) w.ep.push_back(rews[i]);
                else w.ep[i] += rews[i];
            }
            if (done != 0.0 || trunc != 0.0) {
                if (c->avg_none) {
                    c->avg = w.ep[0];
                    c->avg_none = false;
                } else {
                    for (double r : w.ep) c->avg = c->avg * 0.9 + r * 0.1;
                }
                w.ep.assign(1, 0.0);
            }
            bool in = false;
            for (int p : c->current_pids) in |= p == pid;
            if (!in) c->current_pids.push_back(pid);
            w.next_obs = nxt;
            w.next_n = rows;
            w.p_rew.swap(rews);
            w.p_next.swap(nxt);
            w.p_next_n = rows;
            w.p_done = done;
            w.p_trunc = trunc;
            w.has_reward = w.has_next = w.has_done = true;
            if (rows != prev_n) {  // agent count changed across the reset: flush this environment's trajectory, start a fresh assembler
                traj_update(w);
                c->completed.push_back(std::move(w.traj));
                w.traj.clear();
                w.has_state = w.has_reward = w.has_next = w.has_done = false;
            }
            got += prev_n;
        }
    }
    for (int pid : c->current_pids) {
        Worker &w = c->w[pid];
        if (w.next_n >= 0) {
            w.cur_obs.swap(w.next_obs);
            w.cur_n = w.next_n;
            w.next_n = -1;
        }
    }
    for (int pid = 0; pid < c->n; ++pid) {  // _sync_trajectories
        Worker &w = c->w[pid];
        if (traj_update(w)) {
            c->completed.push_back(std::move(w.traj));
            w.traj.clear();
        }
    }
    *n_collected = got;
    return 0;
}

// End of a collect_timesteps call, step 1: every environment's trajectory is flushed (an action in flight keeps its state / action /
// log-prob for the response that arrives during the next call); -> the number of agent-timesteps the flush will emit, the action
// width, the number of metrics records and their total size
int rlppo_collector_finish(void *handle, int64_t *n_steps, int32_t *act_width, int64_t *n_metrics, int64_t *metrics_floats) {
    Collector *c = static_cast<Collector *>(handle);
    if (!c || !n_steps || !act_width || !n_metrics || !metrics_floats) COLLECTOR_FAIL(RLPPO_ERR_ARG, "collector_finish: null argument");
    for (int pid = 0; pid < c->n; ++pid) {
        Worker &w = c->w[pid];
        c->completed.push_back(std::move(w.traj));
        w.traj.clear();
        const bool in_flight = w.has_state && !w.has_reward;
        if (!in_flight) w.has_state = false;
        w.has_reward = w.has_next = w.has_done = false;
    }
    int64_t total = 0;
    for (const auto &t : c->completed)
        if (!t.empty()) total += (int64_t)t[0].rew.size() * (int64_t)t.size();
    *n_steps = total;
    *act_width = c->act_width;
    *n_metrics = (int64_t)c->metrics.size();
    int64_t mf = 0;
    for (const auto &m : c->metrics) mf += (int64_t)m.values.size();
    *metrics_floats = mf;
    return 0;
}

// ... step 2: the trajectories agent by agent (BatchedTrajectory.get_all), the last step of every sequence force-marked truncated if
// it is not done (quirk Q4); rewards / dones / truncated are float64 like the lists the reference builds.  metrics_*: every
// message's metrics record, flat, with its rank and shape (9 ints per record: rank, then up to 8 dimensions).
int rlppo_collector_emit(void *handle, float *states, float *actions, float *log_probs, double *rewards, float *next_states, double *dones,
                         double *truncated, float *metrics_values, int32_t *metrics_shapes) {
    Collector *c = static_cast<Collector *>(handle);
    if (!c) COLLECTOR_FAIL(RLPPO_ERR_ARG, "collector_emit: null handle");
    const int d = c->d, aw = c->act_width;
    int64_t r = 0;
    for (const auto &t : c->completed) {
        if (t.empty()) continue;
        const int agents = (int)t[0].rew.size();
        for (int i = 0; i < agents; ++i) {
            for (size_t k = 0; k < t.size(); ++k) {
                const Step &s = t[k];
                if (i >= s.n_state || i >= (int)s.rew.size() || (int64_t)s.action.size() != (int64_t)s.n_state * aw)
                    COLLECTOR_FAIL(RLPPO_ERR_ARG, "collector_emit: a trajectory's steps disagree about its agents (agent %d of %d, %d state rows, action width %d)", i, agents, s.n_state, aw);
                memcpy(states + r * d, s.state.data() + (size_t)i * d, d * sizeof(float));
                memcpy(actions + r * aw, s.action.data() + (size_t)i * aw, aw * sizeof(float));
                log_probs[r] = s.logp[i];
                rewards[r] = s.rew[i];
                if (i < s.n_next) memcpy(next_states + r * d, s.next.data() + (size_t)i * d, d * sizeof(float));
                else memset(next_states + r * d, 0, d * sizeof(float));  // team size changed across a reset
                dones[r] = s.done;
                truncated[r] = s.trunc;
                ++r;
            }
            truncated[r - 1] = dones[r - 1] == 0.0 ? 1.0 : 0.0;
        }
    }
    c->completed.clear();
    int64_t mo = 0, mi = 0;
    for (const auto &m : c->metrics) {
        if (metrics_values && !m.values.empty()) memcpy(metrics_values + mo, m.values.data(), m.values.size() * sizeof(float));
        mo += (int64_t)m.values.size();
        if (metrics_shapes) {
            metrics_shapes[mi * 9] = (int32_t)m.shape.size();
            for (size_t k = 0; k < 8; ++k) metrics_shapes[mi * 9 + 1 + k] = k < m.shape.size() ? m.shape[k] : 0;
        }
        ++mi;
    }
    c->metrics.clear();
    return 0;
}

// average_reward of the manager (python float or None): get / set (a checkpoint restores it)
int rlppo_collector_average_reward(void *handle, int32_t set, double *value, int32_t *is_none) {
    Collector *c = static_cast<Collector *>(handle);
    if (!c || !value || !is_none) COLLECTOR_FAIL(RLPPO_ERR_ARG, "collector_average_reward: null argument");
    if (set) {
        c->avg_none = *is_none != 0;
        c->avg = *value;
    } else {
        *is_none = c->avg_none ? 1 : 0;
        *value = c->avg;
    }
    return 0;
}

}  // extern "C"
