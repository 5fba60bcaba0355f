// Chains side by side on one device: the rendezvous of their host threads, the shared frozen-factor launches, and
// bgmm_group_sweep_staged.
#include "api_internal.h"

// Chains of one group call that are in the frozen-factor regime TOGETHER (burn-in from a random start) share their
// launches: the hardware runs about four kernels of different streams side by side, whatever the number of streams, so
// eight chains with four small launches per window each queue up behind one another -- while one launch whose grid is
// (x, chain) runs the eight resolvers truly side by side (kernels_gram.hip: *_group_kernel).  The chains' host threads meet
// here.  Every thread declares, once per batch of its sweep loop, either "a batch of frozen-factor windows" (submit: it
// waits) or "something else" (pass: the others do not wait for it); when nobody is undeclared, one of the waiting threads
// is made leader and queues the batch for all waiting chains of its shape on its own stream, behind an event of each
// member's stream; the members' streams wait for the leader's.  Same kernels, same per-chain control blocks: the
// trajectories are those of separate sweeps.
struct GramCombiner {
    enum { UNKNOWN = 0, WAITING = 1, BUSY = 2, DONE = 3 };
    struct Slot { int state = UNKNOWN; bgmm_ctx *c = nullptr; int T = 0; int pipe_T = 0; long long pos = 0; int kind = 0; int result = 0; hipEvent_t ev = nullptr; };
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Slot> slots;
    int leader = -1;
    long long shared_batches = 0, shared_members = 0, piped_batches = 0;

    void elect_locked() {
        if (leader >= 0) return;
        int first = -1;
        for (size_t k = 0; k < slots.size(); ++k) {
            if (slots[k].state == UNKNOWN) return;
            if (slots[k].state == WAITING && first < 0) first = (int)k;
        }
        if (first >= 0) { leader = first; cv.notify_all(); }
    }
    void declare(int i, int state) {
        std::lock_guard<std::mutex> lk(mu);
        slots[(size_t)i].state = state;
        elect_locked();
    }
};

static int gram_group_launch(GramCombiner &G, const std::vector<int> &members, int T, std::vector<hipEvent_t> &ev_of, int kind);
static int gram_group_pipe_launch(GramCombiner &G, const std::vector<int> &members, int T, std::vector<hipEvent_t> &ev_of);

// Returns 0: the batch has been queued with the group's (the chain's stream waits for it), 2: ... as PIPELINED windows;
// 1: queue it yourself; < 0: error.  pipe_T: the pipelined windows this chain could take from visit `pos` on (0: none).
void combiner_declare_busy(bgmm_ctx *c) { c->combiner->declare(c->combiner_slot, GramCombiner::BUSY); }

int combiner_submit(bgmm_ctx *c, int T, int pipe_T, long long pos, int kind) {
    GramCombiner &G = *c->combiner;
    const int me = c->combiner_slot;
    // (a chain that cannot take part queues its batch itself -- and says so, or the others would wait for its declaration)
    if (!c->grp_ev_in) {
        if (hipEventCreateWithFlags(&c->grp_ev_in, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->grp_ev_out, hipEventDisableTiming) != hipSuccess) {
            G.declare(me, GramCombiner::BUSY);
            return 1;
        }
    }
    if (hipEventRecord(c->grp_ev_in, c->stream) != hipSuccess) { G.declare(me, GramCombiner::BUSY); return 1; }
    std::unique_lock<std::mutex> lk(G.mu);
    GramCombiner::Slot &S = G.slots[(size_t)me];
    S.state = GramCombiner::WAITING; S.T = T; S.pipe_T = pipe_T; S.pos = pos; S.kind = kind; S.result = 1; S.ev = nullptr;
    G.elect_locked();
    G.cv.wait(lk, [&] { return S.state != GramCombiner::WAITING || G.leader == me; });
    if (S.state == GramCombiner::WAITING) {
        // leader: the waiting chains of this chain's shape (device, D, column plan)
        std::vector<int> members;
        int Tmax = 0;
        for (size_t k = 0; k < G.slots.size(); ++k) {
            const GramCombiner::Slot &o = G.slots[k];
            if (o.state != GramCombiner::WAITING) continue;
            if (o.kind == kind && o.c->device == c->device && o.c->d.D == c->d.D && o.c->d.gcols == c->d.gcols && o.c->gram_lds == c->gram_lds) {
                members.push_back((int)k);
                if (o.T > Tmax) Tmax = o.T;
            }
        }
        // The members that can take pipelined windows (api_sweep.hip gram_pipe_batch: the next window's cross forms and the
        // last one's finish on a second stream beside the resolvers) get them in shared launches too -- all of them in ONE
        // sequence: the pipeline is what keeps the chip busy through the resolvers there --, the others the plain windows.
        static const bool pipe_on = bgmm_dev_option("group_pipe", 1) != 0;
        std::vector<int> piped, plain;
        int Tpipe = 1 << 30;
        for (int k : members) {
            const GramCombiner::Slot &o = G.slots[(size_t)k];
            if (pipe_on && kind == 0 && o.pipe_T >= 4) { piped.push_back(k); if (o.pipe_T < Tpipe) Tpipe = o.pipe_T; }
            else plain.push_back(k);
        }
        // (the pipeline keeps the chip busy through the resolvers of up to a dozen chains; beyond, the wide kernels of one
        //  launch sequence ARE the window, and two sequences of plain windows whose phases interleave do better -- first sweep
        //  of 12 / 16 / 32 C4-shaped chains from "rand": 7.8 / 8.4 / 9.7 x one chain pipelined, 6.4 / 9.3 / 11.7 x plain)
        if (piped.size() < 2 || members.size() > 12) { plain = members; piped.clear(); }
        int rc_plain = 1, rc_piped = 1;
        std::vector<hipEvent_t> ev_plain, ev_piped;     // per member: the event its stream waits for (its sub-group's)
        lk.unlock();
        if (piped.size() >= 2) rc_piped = gram_group_pipe_launch(G, piped, Tpipe, ev_piped);
        if (plain.size() >= 2) rc_plain = gram_group_launch(G, plain, Tmax, ev_plain, kind);
        lk.lock();
        if (rc_piped == 0) { G.shared_batches += 1; G.piped_batches += 1; G.shared_members += (long long)piped.size(); }
        if (rc_plain == 0) { G.shared_batches += 1; G.shared_members += (long long)plain.size(); }
        // everybody who waited goes on: the members with the shared batch (or, if it could not be queued, on their own),
        // the chains of other shapes on their own
        for (size_t k = 0; k < G.slots.size(); ++k) {
            GramCombiner::Slot &o = G.slots[k];
            if (o.state != GramCombiner::WAITING) continue;
            const auto ip = std::find(piped.begin(), piped.end(), (int)k);
            const auto iq = std::find(plain.begin(), plain.end(), (int)k);
            o.result = 1; o.ev = nullptr;
            if (ip != piped.end()) {
                o.result = rc_piped == 0 ? 2 : rc_piped;
                if (rc_piped == 0) o.ev = ev_piped[(size_t)(ip - piped.begin())];
            } else if (iq != plain.end() && plain.size() >= 2) {
                o.result = rc_plain;
                if (rc_plain == 0) o.ev = ev_plain[(size_t)(iq - plain.begin())];
            }
            o.state = GramCombiner::UNKNOWN;
        }
        G.leader = -1;
        G.cv.notify_all();
    }
    const int result = S.result;
    hipEvent_t ev = S.ev;
    lk.unlock();
    if ((result == 0 || result == 2) && ev && ev != c->grp_ev_out) {       // (a sub-group's leader queued the batch on its own stream)
        // Waited for HERE, on the host, not by the chain's stream: a stream parked at a wait is a hardware queue that sits at
        // a barrier packet for the whole batch, and seven of those beside the two queues the batch runs on cost it a third
        // of its speed (8 C4 chains, first sweep: 6.9 s -> 5.0 s; round 6) -- the command processor keeps coming back to them.
        static const bool host_wait = bgmm_dev_option("group_host_wait", 1) != 0;
        if (host_wait) { if (hipEventSynchronize(ev) != hipSuccess) return BGMM_EDEVICE; }
        else if (hipStreamWaitEvent(c->stream, ev, 0) != hipSuccess) return BGMM_EDEVICE;
    }
    return result;
}

// The shared batch is queued as a few SUB-GROUPS, each on the stream of its first member: the one-workgroup resolvers of
// one sub-group run beside the wide kernels (cross forms, rebuilds) of the others -- with every chain in ONE launch
// sequence the chip idles through each window's resolver phase and the resolvers wait through its wide phases (eight
// C4 chains: 201 + 132 us per window whatever runs beside them).  One stream per chain, the other extreme, keeps only
// about four kernels in flight and stretches every one of them.  BGMM_DEV_OPTIONS group_split
// overrides the number of sub-groups (1: one launch sequence for all).
static int gram_group_launch(GramCombiner &G, const std::vector<int> &members, int T, std::vector<hipEvent_t> &ev_of, int kind) {
    bgmm_ctx *lead = G.slots[(size_t)members[0]].c;
    const int m = (int)members.size();
    ev_of.assign((size_t)m, nullptr);
    if (hipSetDevice(lead->device) != hipSuccess) return 1;
    if (lead->grp_devs_cap < m) {
        if (lead->grp_devs) (void)hipFree(lead->grp_devs);
        lead->grp_devs = nullptr; lead->grp_devs_cap = 0;
        if (hipMalloc((void **)&lead->grp_devs, sizeof(Dev) * (size_t)m) != hipSuccess) return 1;
        lead->grp_devs_cap = m;
    }
    static const int split_env = bgmm_dev_option("group_split", 0);
    // (safe-stay steps keep a second stream busy themselves -- the look-ahead --: one sequence for all chains measured best,
    //  8 C4-shaped chains 14.9 sweeps/s against 13.5 with two sub-groups and 11.5 with four)
    int n_sub = split_env > 0 ? split_env : (kind == 1 ? 1 : (m >= 4 ? 2 : 1));
    if (n_sub > m / 2) n_sub = m / 2 > 0 ? m / 2 : 1;
    std::vector<Dev> views((size_t)m);
    std::vector<int> reach_of((size_t)n_sub, 0), lo_of((size_t)n_sub + 1, 0), nslots_of((size_t)n_sub, 0);
    std::vector<long long> rows_of((size_t)n_sub, 0);
    for (int g = 0; g <= n_sub; ++g) lo_of[(size_t)g] = (int)((long long)m * g / n_sub);
    for (int g = 0; g < n_sub; ++g) {
        bgmm_ctx *sl = G.slots[(size_t)members[(size_t)lo_of[(size_t)g]]].c;
        for (int k = lo_of[(size_t)g]; k < lo_of[(size_t)g + 1]; ++k) {
            bgmm_ctx *o = G.slots[(size_t)members[(size_t)k]].c;
            views[(size_t)k] = o->d;
            const int r = o->d.gram_K + o->d.gram_terms / 2 + 2 + 32;
            if (r > reach_of[(size_t)g]) reach_of[(size_t)g] = r;
            if (o->d.nslots > nslots_of[(size_t)g]) nslots_of[(size_t)g] = o->d.nslots;
            if (o->d.batch_rows > rows_of[(size_t)g]) rows_of[(size_t)g] = o->d.batch_rows;
            // (what the member has queued on its own stream -- the sweep's opening, rebuilt factors -- comes first)
            if (o != sl && hipStreamWaitEvent(sl->stream, o->grp_ev_in, 0) != hipSuccess) return 1;
            ev_of[(size_t)k] = sl->grp_ev_out;
        }
    }
    // (safe-stay steps: a sub-group takes the look-ahead's route iff every chain in it would -- the same chunk for all; its
    //  second stream and events are its first chain's)
    std::vector<int> ahead_of((size_t)n_sub, 0);
    if (kind == 1)
        for (int g = 0; g < n_sub; ++g) {
            bgmm_ctx *sl = G.slots[(size_t)members[(size_t)lo_of[(size_t)g]]].c;
            long long C = views[(size_t)lo_of[(size_t)g]].ahead_C;
            for (int k = lo_of[(size_t)g]; k < lo_of[(size_t)g + 1]; ++k)
                if (views[(size_t)k].ahead_C != C) C = 0;
            if (C > 0 && !sl->ahead_stream) {
                if (hipStreamCreateWithFlags(&sl->ahead_stream, hipStreamNonBlocking) != hipSuccess) C = 0;
                else
                    for (auto &row : sl->ahead_ev)
                        for (hipEvent_t &e : row)
                            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return 1;
            }
            ahead_of[(size_t)g] = C > 0 ? 1 : 0;
            static const bool lazy = bgmm_dev_option("group_lazy", 0) != 0;
            for (int k = lo_of[(size_t)g]; k < lo_of[(size_t)g + 1]; ++k) {
                if (C <= 0) views[(size_t)k].ahead_C = 0;
                views[(size_t)k].ahead_lazy = (C > 0 && lazy) ? 1 : 0;
            }
        }
    // (a blocking copy: the views are host memory of this call; the array's last readers -- the shared batch before this one --
    // have been waited for by every one of its members)
    if (hipMemcpy(lead->grp_devs, views.data(), sizeof(Dev) * (size_t)m, hipMemcpyHostToDevice) != hipSuccess) return 1;
    // (from here on a failure is an error for every member, not a reason to queue their batches separately: part of the shared
    // batch may already be in the queue, and separate launches would run beside it on the same chains)
    // window by window across the sub-groups, so that the host queues them at the same pace
    if (kind == 1)
        for (int g = 0; g < n_sub; ++g) {
            bgmm_ctx *sl = G.slots[(size_t)members[(size_t)lo_of[(size_t)g]]].c;
            launch_safe_open_group(lead->grp_devs + lo_of[(size_t)g], lo_of[(size_t)g + 1] - lo_of[(size_t)g], sl->stream);
        }
    for (int t = 0; t < T; ++t)
        for (int g = 0; g < n_sub; ++g) {
            bgmm_ctx *sl = G.slots[(size_t)members[(size_t)lo_of[(size_t)g]]].c;
            const Dev *grp = lead->grp_devs + lo_of[(size_t)g];
            const int cnt = lo_of[(size_t)g + 1] - lo_of[(size_t)g];
            bool ok;
            if (kind == 1) {
                // (a safe-stay step -- the touched labels of every chain's open stretch re-scored, its verdicts and list, then
                //  the frozen-factor kernels on the listed rows; the look-ahead's ring kept current on the sub-group's second
                //  stream: launch_safe_step, kernels_safe.hip)
                const bool ah_on = ahead_of[(size_t)g] != 0;
                // (the look-ahead's stream: the sub-group's SECOND chain's own -- idle while its thread waits, and, created right
                //  behind the first chain's, on another hardware queue; the first chain's own look-ahead stream, created much
                //  later, shared the main stream's queue in one process out of four: 13.6 instead of 16.9 sweeps/s for eight chains)
                hipStream_t second = cnt >= 2 ? G.slots[(size_t)members[(size_t)lo_of[(size_t)g] + 1]].c->stream : sl->ahead_stream;
                SafeAhead ah{second, sl->ahead_ev[0][t & 7], sl->ahead_ev[1][t & 7]};
                if (ah_on && t > 0 && hipStreamWaitEvent(sl->stream, sl->ahead_ev[1][(t - 1) & 7], 0) != hipSuccess) return BGMM_EDEVICE;
                ok = launch_safe_group_step(views[(size_t)lo_of[(size_t)g]], grp, cnt, reach_of[(size_t)g], sl->gram_lds, rows_of[(size_t)g],
                                            nslots_of[(size_t)g], sl->stream, ah_on ? &ah : nullptr);
            } else {
                ok = launch_gram_group_step(sl->d, grp, cnt, reach_of[(size_t)g], sl->gram_lds, sl->stream);
            }
            if (!ok) return BGMM_EDEVICE;
        }
    if (kind == 1)
        for (int g = 0; g < n_sub; ++g) {
            bgmm_ctx *sl = G.slots[(size_t)members[(size_t)lo_of[(size_t)g]]].c;       // (the second stream is idle when the batch ends)
            if (ahead_of[(size_t)g] && hipStreamWaitEvent(sl->stream, sl->ahead_ev[1][(T - 1) & 7], 0) != hipSuccess) return BGMM_EDEVICE;
        }
    if (hipGetLastError() != hipSuccess) return BGMM_EDEVICE;
    for (int g = 0; g < n_sub; ++g) {
        bgmm_ctx *sl = G.slots[(size_t)members[(size_t)lo_of[(size_t)g]]].c;
        if (hipEventRecord(sl->grp_ev_out, sl->stream) != hipSuccess) return BGMM_EDEVICE;
    }
    return 0;
}

// A batch of T PIPELINED frozen-factor windows for the chains `members` (>= 2, one shape), every kernel of gram_pipe_batch's
// schedule (api_sweep.hip) ONE launch for all of them -- workgroup (x, chain) --, on the first member's two streams:
//   main stream   cross(0)  resolve(0)  carry(1) resolve(1)  carry(2) resolve(2) ...
//   second stream     cross(1)      finish(0) cross(2)   finish(1) cross(3) ...
// Chain c's window k starts at its own visit pos_c + 64 k and works in its buffer set k & 1 (gram_pgroup_view).  A chain
// whose pipeline breaks (Ctrl::pipe_break: a component opened or deleted, a window that ended early) stands still for the rest
// of the batch -- its workgroups return at once -- and the others carry on; each host thread reads its own control block.
static int gram_group_pipe_launch(GramCombiner &G, const std::vector<int> &members, int T, std::vector<hipEvent_t> &ev_of) {
    bgmm_ctx *lead = G.slots[(size_t)members[0]].c;
    const int m = (int)members.size();
    ev_of.assign((size_t)m, nullptr);
    if (hipSetDevice(lead->device) != hipSuccess) return 1;
    if (lead->grp_pdevs_cap < 2 * m) {
        if (lead->grp_pdevs) (void)hipFree(lead->grp_pdevs);
        lead->grp_pdevs = nullptr; lead->grp_pdevs_cap = 0;
        if (hipMalloc((void **)&lead->grp_pdevs, sizeof(Dev) * 2 * (size_t)m) != hipSuccess) return 1;
        lead->grp_pdevs_cap = 2 * m;
    }
    while (lead->pipe_ev.size() < (size_t)(3 * T + 5)) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return 1;
        lead->pipe_ev.push_back(e);
    }
    // The second stream is the SECOND MEMBER's own: idle while its thread waits here, and what that chain queues later comes
    // behind the batch anyway (a stream of the leader's own, or raised priorities for the chains' main streams, measured the
    // same: 4.74 - 4.78 s for 8 C4 chains' first sweep).
    hipStream_t M = lead->stream, S = G.slots[(size_t)members[1]].c->stream;
    std::vector<Dev> views(2 * (size_t)m);
    int reach = 0, max_K = 0;
    for (int k = 0; k < m; ++k) {
        const GramCombiner::Slot &sl = G.slots[(size_t)members[(size_t)k]];
        bgmm_ctx *o = sl.c;
        for (int p = 0; p < 2; ++p) {
            Dev &v = views[(size_t)p * (size_t)m + (size_t)k];
            v = o->d;
            gram_point_view(o, v, p);
            v.pipe = 1;
            v.pipe_pos = sl.pos;
        }
        views[(size_t)k].xp_in = views[(size_t)m + (size_t)k].xp_out;          // (what the window before exported: the other set's)
        views[(size_t)m + (size_t)k].xp_in = views[(size_t)k].xp_out;
        const int r = o->d.gram_K + o->d.gram_terms / 2 + 2 + 32;
        if (r > reach) reach = r;
        if (o->d.gram_K > max_K) max_K = o->d.gram_K;
        // (what the member has queued on its own stream -- the sweep's opening, rebuilt factors -- comes first)
        if (o != lead && hipStreamWaitEvent(M, o->grp_ev_in, 0) != hipSuccess) return 1;
        ev_of[(size_t)k] = lead->grp_ev_out;
    }
    // (a blocking copy: the views are host memory of this call; the array's last readers -- the shared batch before this one --
    // have been waited for by every one of its members)
    if (hipMemcpy(lead->grp_pdevs, views.data(), sizeof(Dev) * 2 * (size_t)m, hipMemcpyHostToDevice) != hipSuccess) return 1;
    const Dev *v0 = lead->grp_pdevs, *v1 = lead->grp_pdevs + m;
    // (from here on a failure is an error for every member: part of the shared batch may already be in the queue)
    for (int k = 0; k < m; ++k)
        if (hipMemsetAsync(&G.slots[(size_t)members[(size_t)k]].c->d.ctrl->pipe_break, 0, sizeof(int), M) != hipSuccess) return BGMM_EDEVICE;
    auto evG = [&](int k) { return lead->pipe_ev[(size_t)(2 + 3 * k)]; };
    auto evR = [&](int k) { return lead->pipe_ev[(size_t)(3 + 3 * k)]; };
    auto evC = [&](int k) { return lead->pipe_ev[(size_t)(4 + 3 * k)]; };
    const Dev &L = lead->d;
    // (finish(k - 1) is held back until carry(k) is through: the two start at the same moment otherwise -- the end of
    //  resolve(k - 1) --, and the carry, which is on the chain of every window, took 66 us beside the finish of eight chains
    //  against 14 alone.  BGMM_DEV_OPTIONS="group_carry_first=0": gram_pipe_batch's order)
    static const bool carry_first = bgmm_dev_option("group_carry_first", 1) != 0;
    if (!launch_gram_cross_pgroup(L, v0, v1, m, 0, false, max_K, M)) return BGMM_EDEVICE;
    if (hipEventRecord(lead->pipe_ev[0], M) != hipSuccess || hipStreamWaitEvent(S, lead->pipe_ev[0], 0) != hipSuccess) return BGMM_EDEVICE;
    if (T > 1) {
        launch_gram_cross_pgroup(L, v0, v1, m, 1, true, max_K, S);
        if (hipEventRecord(evG(1), S) != hipSuccess) return BGMM_EDEVICE;
    }
    for (int k = 0; k < T; ++k) {
        if (k > 0) {
            if (hipStreamWaitEvent(M, evG(k), 0) != hipSuccess) return BGMM_EDEVICE;
            launch_gram_carry_pgroup(v0, v1, m, k, M);
            if (carry_first) {
                // the second stream's share of window k - 1, behind this carry
                if (hipEventRecord(evC(k), M) != hipSuccess || hipStreamWaitEvent(S, evR(k - 1), 0) != hipSuccess ||
                    hipStreamWaitEvent(S, evC(k), 0) != hipSuccess) return BGMM_EDEVICE;
                launch_gram_finish_pgroup(L, v0, v1, m, k - 1, S);
                if (k + 1 < T) {
                    launch_gram_cross_pgroup(L, v0, v1, m, k + 1, true, max_K, S);
                    if (hipEventRecord(evG(k + 1), S) != hipSuccess) return BGMM_EDEVICE;
                }
            }
        }
        launch_gram_resolve_pgroup(L, v0, v1, m, k, reach, lead->gram_lds, M);
        if (hipEventRecord(evR(k), M) != hipSuccess) return BGMM_EDEVICE;
        if (!carry_first) {
            if (hipStreamWaitEvent(S, evR(k), 0) != hipSuccess) return BGMM_EDEVICE;
            launch_gram_finish_pgroup(L, v0, v1, m, k, S);
            if (k + 2 < T) {
                launch_gram_cross_pgroup(L, v0, v1, m, k + 2, true, max_K, S);
                if (hipEventRecord(evG(k + 2), S) != hipSuccess) return BGMM_EDEVICE;
            }
        }
    }
    if (carry_first) {
        if (hipStreamWaitEvent(S, evR(T - 1), 0) != hipSuccess) return BGMM_EDEVICE;
        launch_gram_finish_pgroup(L, v0, v1, m, T - 1, S);
    }
    if (hipEventRecord(lead->pipe_ev[1], S) != hipSuccess || hipStreamWaitEvent(M, lead->pipe_ev[1], 0) != hipSuccess) return BGMM_EDEVICE;
    if (hipGetLastError() != hipSuccess) return BGMM_EDEVICE;
    if (hipEventRecord(lead->grp_ev_out, M) != hipSuccess) return BGMM_EDEVICE;
    return 0;
}

// Sweeps of several chains that live on ONE device, side by side.  Chains that can take the one-workgroup sweep (D <= 4,
// full covariance, automatic tuning, labels within the LDS plan) are opened and swept by two launches for all of them
// -- one workgroup, one compute unit per chain -- instead of two launches and a host round trip each; every other chain
// is swept on its own as bgmm_sweep_staged would.  Same trajectories as separate calls.
extern "C" int bgmm_group_sweep_staged(bgmm_ctx *const *ctxs, int32_t n, const int32_t *use_power, const double *power,
                                       int32_t *rc_out) {
    if (!ctxs || n < 1 || !rc_out) return BGMM_EINVAL;
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return BGMM_EINVAL;
        rc_out[i] = 0;
        for (int j = 0; j < i; ++j)
            if (ctxs[j] == ctxs[i]) return fail(ctxs[i], BGMM_EINVAL, "a context appears twice in the group");
    }
    std::vector<int> deferred;
    int worst = 0;
    // Chains that can never take the one-workgroup sweep (D > 4, diagonal / fixed covariance) run their whole sweep as
    // bgmm_sweep_staged would -- but CONCURRENTLY, each on its own stream, each driven by its own host thread (a context is
    // one host thread's at a time; distinct contexts share nothing).  What bounds such a sweep while the chain still moves is
    // a latency chain that keeps ONE workgroup busy (kernels_gram.hip: the resolver); G chains side by side keep G of them
    // busy, and the wide kernels of one chain (cross forms, rebuilds) run beside the resolvers of the others.
    std::vector<std::thread> workers;
    std::vector<int> threaded;
    for (int i = 0; i < n; ++i)
        if (!seq_shape(ctxs[i])) threaded.push_back(i);
    GramCombiner comb;
    if (threaded.size() >= 2) {
        comb.slots.resize(threaded.size());
        for (size_t k = 0; k < threaded.size(); ++k) {
            comb.slots[k].c = ctxs[threaded[k]];
            ctxs[threaded[k]]->combiner_slot = (int)k;
        }
        // (std::thread's constructor may throw -- no exception may cross the C ABI, least of all with joinable workers
        //  behind it: the chains whose thread could not be started are declared DONE for the rendezvous, so that nobody
        //  waits for them, and swept on this thread)
        size_t started = 0;
        try {
            workers.reserve(threaded.size());
            for (int i : threaded) {
                const int up = use_power ? use_power[i] : 0;
                const double pw = (up && power) ? power[i] : 1.0;
                workers.emplace_back([=, &comb]() {
                    ctxs[i]->combiner = &comb;
                    rc_out[i] = sweep_impl(ctxs[i], up, pw, 0);
                    comb.declare(ctxs[i]->combiner_slot, GramCombiner::DONE);
                    ctxs[i]->combiner = nullptr;
                });
                ++started;
            }
        } catch (...) {
        }
        for (size_t k = started; k < threaded.size(); ++k) comb.declare((int)k, GramCombiner::DONE);
        for (size_t k = started; k < threaded.size(); ++k) {
            const int i = threaded[k];
            const int up = use_power ? use_power[i] : 0;
            rc_out[i] = sweep_impl(ctxs[i], up, (up && power) ? power[i] : 1.0, 0);
        }
    } else {
        threaded.clear();
    }
    for (int i = 0; i < n; ++i) {
        if (std::find(threaded.begin(), threaded.end(), i) != threaded.end()) continue;
        const int up = use_power ? use_power[i] : 0;
        const int rc = sweep_impl(ctxs[i], up, (up && power) ? power[i] : 1.0, 1);
        if (rc == 1) deferred.push_back(i);
        else { rc_out[i] = rc; if (rc < 0 && worst == 0) worst = rc; }
    }
    // one pair of launches per (device, D, LDS plan) among the chains that wait
    std::vector<char> done(deferred.size(), 0);
    for (size_t a = 0; a < deferred.size(); ++a) {
        if (done[a]) continue;
        bgmm_ctx *lead = ctxs[deferred[a]];
        std::vector<int> grp;
        for (size_t b = a; b < deferred.size(); ++b) {
            bgmm_ctx *o = ctxs[deferred[b]];
            if (!done[b] && o->device == lead->device && o->d.D == lead->d.D && o->grp_cap == lead->grp_cap) {
                grp.push_back(deferred[b]);
                done[b] = 1;
            }
        }
        const int m = (int)grp.size();
        hipError_t e = hipSetDevice(lead->device);
        if (e == hipSuccess && lead->grp_devs_cap < m) {
            if (lead->grp_devs) (void)hipFree(lead->grp_devs);
            lead->grp_devs = nullptr; lead->grp_devs_cap = 0;
            e = hipMalloc((void **)&lead->grp_devs, sizeof(Dev) * (size_t)m);
            if (e == hipSuccess) lead->grp_devs_cap = m;
        }
        std::vector<Dev> views((size_t)m);
        for (int k = 0; k < m && e == hipSuccess; ++k) {
            views[(size_t)k] = ctxs[grp[(size_t)k]]->d;
            // (what phase 1 queued on the chain's own stream -- a new seating table, stale factors rebuilt -- has to be there)
            e = hipStreamSynchronize(ctxs[grp[(size_t)k]]->stream);
        }
        hipStream_t st = lead->stream;
        if (e == hipSuccess) e = hipMemcpyAsync(lead->grp_devs, views.data(), sizeof(Dev) * (size_t)m, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            launch_sweep_begin(lead->d, st, lead->grp_devs, m);
            if (!launch_sweep_seq(lead->d, lead->grp_cap, st, lead->grp_devs, m)) e = hipErrorLaunchFailure;
        }
        if (e == hipSuccess) e = hipGetLastError();
        for (int k = 0; k < m && e == hipSuccess; ++k) {
            bgmm_ctx *o = ctxs[grp[(size_t)k]];
            e = hipMemcpyAsync(o->ctrl_host, o->d.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, st);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);      // (views is pageable: the copy above has been staged by now)
        for (int k = 0; k < m; ++k) {
            bgmm_ctx *o = ctxs[grp[(size_t)k]];
            int rc;
            if (e != hipSuccess) {
                o->err = std::string("group sweep: ") + hipGetErrorString(e);
                rc = BGMM_EDEVICE;
            } else {
                rc = sweep_impl(o, o->d.use_power, o->d.power, 2);
            }
            rc_out[grp[(size_t)k]] = rc;
            if (rc < 0 && worst == 0) worst = rc;
        }
    }
    for (auto &w : workers) w.join();
    for (int i : threaded)
        if (rc_out[i] < 0 && worst == 0) worst = rc_out[i];
    return worst;
}
