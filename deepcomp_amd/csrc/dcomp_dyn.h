// dcomp_dyn.h -- step kernel for envs whose UE list changes during the episode (UE arrival / departure,
// base.py:433-443, 592-618).  Included by dcomp_device.h.  Same pipeline as step_kernel (generic sharing table), plus
// an event phase between the actions and the rates:
//   * an env has U = max_ues slots (base.py:79-84); slot = position in the reference's env.ue_list, which is the order
//     central observations use; `uid[slot]` carries the UE id (bit 15: arrived during the episode);
//   * departure: ue_list.pop(idx) -- slots behind idx move up by one (one DPP-free __shfl_down per state word inside
//     the env's lane group), the vacated slot dies; the index comes from the reference's global random.randint (tape
//     mode, drawn on the host) or from a keyed Philox draw;
//   * arrival: the first free slot becomes a UE at a border point of the map (map.py:52-65), id = last id + 1,
//     velocity 'slow', freshly seeded movement stream (base.py:592-606);
//   * the number of UEs is the same in every env (the schedule is configuration), so it is a kernel argument; dead
//     slots write zero rows (the reference zero-pads central observations, central.py:46-55).
// max-cap BSs: the per-slot step-of-connection rows shift with the slots.  Envs wider than one wavefront (max_ues > 64)
// exchange the slot state through LDS instead.
#pragma once

namespace dcomp {

template <int UPAD, class T>
__device__ __forceinline__ T shift_up(bool take, T mine, T next)
{
    return take ? next : mine;
}

// Envs wider than a wavefront exchange the slot state through LDS (word-major: conflict-free) instead of __shfl_down.
template <int UPAD, int NWORDS>
struct DynExchange { uint32_t w[UPAD > 64 ? NWORDS * DCOMP_BLOCK : 1]; };

template <int B, int UPAD>
__global__ __launch_bounds__(DCOMP_BLOCK) void step_kernel_dyn(const KParams p)
{
    using G = Geo<B, UPAD>;
    constexpr int MP = MP_GENERIC;
    constexpr int CSW = (B + 1) / 2, XW = 9 + CSW;
    __shared__ BlockSharedT<B, UPAD> sh;
    __shared__ DynExchange<UPAD, XW> dx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int env_local = tid / UPAD, u = tid % UPAD;
    const int env = blockIdx.x * G::GPB + env_local;
    const bool active = (env < p.E) && (u < p.U);            // owns a slot
    const int idx = env * p.U + u;
    const int gbase = lane & ~(G::WG - 1);
    int cur = p.cur_ue;                                        // UEs in the list before this step's events
    bool alive = active && u < cur;

    double px = 0.0, py = 0.0;
    unsigned long long mv = 0;
    uint32_t conn = 0, act = 0, uidw = 0;
    float ewma = 0.f;
    if (alive) {
        double2 q = p.pos[idx];
        px = q.x; py = q.y;
        mv = p.mv[idx];
        conn = p.conn[idx];
        ewma = p.ewma[idx];
        act = p.action[idx];
        uidw = p.uid[idx];
    }

    // max-cap BSs: the step-of-connection row of this slot (dcomp_state.conn_since) travels with the UE when slots shift.
    // Kept in registers (u16 pairs) from here to the write-back before the rates; every lane reads / writes only its own row.
    uint32_t cs[CSW];
    const bool mc = p.any_maxcap != 0;                         // uniform
    if (mc) {
#pragma unroll
        for (int j = 0; j < CSW; j++) {
            uint32_t lo = 0, hi = 0;
            if (alive) {
                lo = p.conn_since[(size_t)idx * B + 2 * j];
                if (2 * j + 1 < B) hi = p.conn_since[(size_t)idx * B + 2 * j + 1];
            }
            cs[j] = lo | (hi << 16);
        }
    }

    // 1./2. pairs at the pre-move position, toggle (base.py:247-263)
    float l2[B];
    uint32_t in_range = eval_pairs<B>(px, py, p, l2);
    if (act > (uint32_t)B) { atomicOr(p.flags, DCOMP_FLAG_BAD_ACTION); act = 0; }
    if (act > 0) {
        const uint32_t bit = 1u << (act - 1);
        if (conn & bit) conn &= ~bit;
        else if (in_range & bit) {
            conn |= bit;
            if (mc && (p.maxcap_mask & bit)) {                  // connection order = (step, slot): see shared_rates
                const int b = (int)act - 1;
#pragma unroll
                for (int j = 0; j < CSW; j++)
                    if (j == b / 2) cs[j] = (b & 1) ? ((cs[j] & 0x0000FFFFu) | (p.time << 16)) : ((cs[j] & 0xFFFF0000u) | (p.time & 0xFFFFu));
            }
        }
    }

    // 2b. UE departure / arrival (base.py:433-443), after the actions and before the rates
    if (p.n_remove > 0 || p.n_add > 0) {
        for (int k = 0; k < p.n_remove; k++) {                 // base.py:608-618: pop(idx) + disconnect_from_all
            int r;
            if (p.rng_mode == DCOMP_RNG_TAPE) r = (env < p.E) ? p.ev_remove[(size_t)env * p.n_remove + k] : 0;
            else {
                uint32_t d[4];
                philox4x32_10(p.env_base + (uint32_t)env, 0xFFFE0000u + p.ev_rem_base + (uint32_t)k, p.episode, 0u, p.seed_lo, p.seed_hi, d);
                r = (int)__umulhi(d[0], (uint32_t)cur);
            }
            if (alive && u == r && !(uidw & UID_BORN) && p.orig_consumed)      // host bookkeeping of the initial UEs' streams
                p.orig_consumed[(size_t)env * p.U0 + ((uidw & 0x7FFFu) - 1u)] = (uint16_t)(mv >> 48);
            const bool take = u >= r && u + 1 < cur;            // slots behind the leaver move up
            if constexpr (UPAD <= 64) {
                const double nx = __shfl_down(px, 1, UPAD), ny = __shfl_down(py, 1, UPAD);
                const unsigned long long nmv = __shfl_down(mv, 1, UPAD);
                const uint32_t nconn = __shfl_down(conn, 1, UPAD), nuid = __shfl_down(uidw, 1, UPAD);
                const float newma = __shfl_down(ewma, 1, UPAD);
                px = shift_up<UPAD>(take, px, nx); py = shift_up<UPAD>(take, py, ny); mv = shift_up<UPAD>(take, mv, nmv);
                conn = shift_up<UPAD>(take, conn, nconn); uidw = shift_up<UPAD>(take, uidw, nuid); ewma = shift_up<UPAD>(take, ewma, newma);
                if (mc) {
#pragma unroll
                    for (int j = 0; j < CSW; j++) { const uint32_t n = __shfl_down(cs[j], 1, UPAD); cs[j] = take ? n : cs[j]; }
                }
            } else {                                            // the env spans several wavefronts of this workgroup
                uint32_t wd[XW];
                const unsigned long long bx = (unsigned long long)__double_as_longlong(px), by = (unsigned long long)__double_as_longlong(py);
                wd[0] = (uint32_t)bx; wd[1] = (uint32_t)(bx >> 32); wd[2] = (uint32_t)by; wd[3] = (uint32_t)(by >> 32);
                wd[4] = (uint32_t)mv; wd[5] = (uint32_t)(mv >> 32); wd[6] = conn; wd[7] = uidw; wd[8] = __float_as_uint(ewma);
#pragma unroll
                for (int j = 0; j < CSW; j++) wd[9 + j] = mc ? cs[j] : 0u;
#pragma unroll
                for (int w = 0; w < XW; w++) dx.w[w * DCOMP_BLOCK + tid] = wd[w];
                __syncthreads();
                const int src = tid + 1 < DCOMP_BLOCK ? tid + 1 : tid;   // take implies the next slot belongs to the same env
                if (take) {
#pragma unroll
                    for (int w = 0; w < XW; w++) wd[w] = dx.w[w * DCOMP_BLOCK + src];
                }
                __syncthreads();
                px = __longlong_as_double((long long)(((unsigned long long)wd[1] << 32) | wd[0]));
                py = __longlong_as_double((long long)(((unsigned long long)wd[3] << 32) | wd[2]));
                mv = ((unsigned long long)wd[5] << 32) | wd[4];
                conn = wd[6]; uidw = wd[7]; ewma = __uint_as_float(wd[8]);
                if (mc) {
#pragma unroll
                    for (int j = 0; j < CSW; j++) cs[j] = wd[9 + j];
                }
            }
            cur -= 1;
            if (u == cur) { conn = 0; uidw = 0; ewma = 0.f; mv = 0; px = 0.0; py = 0.0; if (mc) { for (int j = 0; j < CSW; j++) cs[j] = 0; } }
            alive = active && u < cur;
        }
        for (int k = 0; k < p.n_add; k++) {                    // base.py:592-606
            uint32_t last;                                                       // id of ue_list[-1]
            if constexpr (UPAD <= 64) last = __shfl(uidw, gbase + (cur - 1), 64);
            else {
                dx.w[tid] = uidw;
                __syncthreads();
                last = dx.w[env_local * UPAD + (cur - 1)];
                __syncthreads();
            }
            if (active && u == cur) {
                int x, y;
                if (p.rng_mode == DCOMP_RNG_TAPE) { const size_t t = ((size_t)env * p.n_add + k) * 2; x = p.ev_add_xy[t]; y = p.ev_add_xy[t + 1]; }
                else {                                          // map.rand_border_point (map.py:52-65)
                    uint32_t d[4];
                    philox4x32_10(p.env_base + (uint32_t)env, 0xFFFF0000u + p.ev_add_base + (uint32_t)k, p.episode, 0u, p.seed_lo, p.seed_hi, d);
                    const int rx = (int)__umulhi(d[0], (uint32_t)p.map_w + 1u), ry = (int)__umulhi(d[1], (uint32_t)p.map_h + 1u);
                    const int border = (int)__umulhi(d[2], 4u);            // left, right, top, bottom
                    x = border == 0 ? 0 : border == 1 ? p.map_w - 1 : rx;
                    y = border == 2 ? p.map_h - 1 : border == 3 ? 0 : ry;
                }
                uidw = ((last & 0x7FFFu) + 1u) | UID_BORN;
                px = (double)x; py = (double)y;
                uint32_t vel, wx, wy;
                draw_triple(p, env, uidw, 0u, p.episode, vel, wx, wy, load_mv_cfg(p, uidw));
                mv = mv_pack(wx, wy, vel, 0u, 0u, 1u);
                conn = 0; ewma = 0.f;
                if (mc) { for (int j = 0; j < CSW; j++) cs[j] = 0; }
            }
            cur += 1;
        }
        alive = active && u < cur;
        in_range = eval_pairs<B>(px, py, p, l2);                // slots changed owners
    }
    if (mc && active) {                                         // rows back in place before shared_rates reads its own
#pragma unroll
        for (int j = 0; j < CSW; j++) {
            p.conn_since[(size_t)idx * B + 2 * j] = (uint16_t)(cs[j] & 0xFFFFu);
            if (2 * j + 1 < B) p.conn_since[(size_t)idx * B + 2 * j + 1] = (uint16_t)(cs[j] >> 16);
        }
    }
    const uint32_t id0 = (uidw & 0x7FFFu) - 1u;
    bool step_util = false;
    float dr_req = 1.f;
    if (alive && !(uidw & UID_BORN) && !p.all_log_util) { UeCfg c = p.ue_cfg[id0]; step_util = c.util == DCOMP_UTIL_STEP; dr_req = c.dr_req; }

    // 3. rates before the move -> reward_before
    float dr[B], cnt[B];
    shared_rates<B, UPAD, MP>(p, sh, conn, l2, ewma, px, py, u, idx, env_local, wave, lane, gbase, dr, cnt);
    float curr = 0.f;
#pragma unroll
    for (int b = 0; b < B; b++) curr += dr[b];
    const float util_pre = ue_utility(curr, step_util, dr_req);
    const float reward_before = clamp_med3(util_pre, MIN_UTIL, MAX_UTIL) * (1.0f / MAX_UTIL);
    // 4. move
    if (alive) {
        move_ue(p, env, uidw, p.episode, px, py, mv, load_mv_cfg(p, uidw));
        if (px < 0.0 || py < 0.0 || px > (double)p.map_w || py > (double)p.map_h) atomicOr(p.flags, DCOMP_FLAG_OUTSIDE_MAP);
    }
    // 5. drop + EWMA
    in_range = eval_pairs<B>(px, py, p, l2);
    conn &= in_range;
    float stale = 0.f;
#pragma unroll
    for (int b = 0; b < B; b++) stale += ((conn >> b) & 1u) ? dr[b] : 0.f;
    ewma = __builtin_fmaf(0.9f, stale, 0.1f * ewma);   // one explicit contraction: every kernel variant rounds alike
    // 6. rates after the move
    shared_rates<B, UPAD, MP>(p, sh, conn, l2, ewma, px, py, u, idx, env_local, wave, lane, gbase, dr, cnt);
    curr = 0.f;
#pragma unroll
    for (int b = 0; b < B; b++) curr += dr[b];
    const float util = ue_utility(curr, step_util, dr_req);
    // 7. state write-back: every slot (dead slots are cleared)
    if (active) {
        p.pos[idx] = alive ? make_double2(px, py) : make_double2(0.0, 0.0);
        p.mv[idx] = alive ? mv : 0ull;
        p.conn[idx] = alive ? conn : 0u;
        p.ewma[idx] = alive ? ewma : 0.f;
        p.uid[idx] = alive ? (uint16_t)uidw : (uint16_t)0;
    }
    // 8. observation, reward, info
    const Outs o{p.obs, p.reward, p.sum_util, p.ue_dr, p.ue_util, p.rb_out, p.next_act, p.obs_compact};
    write_outputs<B, UPAD, false, true>(p, o, sh, active, env, env_local, u, idx, wave, lane, gbase, alive ? conn : 0u, in_range, l2, cnt, util,
                                  curr, reward_before, alive, cur);
}

}  // namespace dcomp
