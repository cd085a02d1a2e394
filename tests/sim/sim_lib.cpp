// TEST INFRASTRUCTURE -- builds tests/_build/libaecm_sim.so: the product's block-DSP source
// (webrtc_aecm_amd/csrc/aecm_wave.h) instantiated on the 64-lane CPU simulator policy.
#include <stdint.h>
#include <string.h>

#include <vector>

#include "wave_sim.h"
#include "aecm_host_state.h"
#include "aecm_wave.h"

using namespace aecm;

struct SimStream {
    StreamImage img;
    std::vector<uint16_t> hist;
    SimStream() : hist(kHistWordsPerStream, 0) {}
};

extern "C" {

void *sim_create(int fs, int cng_mode, int echo_mode) {
    SimStream *s = new SimStream();
    if (!BuildInitImage(fs, &s->img) || !ApplyConfig(s->img.scal.data(), cng_mode, echo_mode)) {
        delete s;
        return nullptr;
    }
    return s;
}
void sim_free(void *h) { delete (SimStream *)h; }
void sim_control(void *h, int fixed_delay, int nlp_flag) { ApplyControl(((SimStream *)h)->img.scal.data(), fixed_delay, nlp_flag); }
void sim_set_echo_path(void *h, const int16_t *path) {
    SimStream *s = (SimStream *)h;
    SetEchoPath(s->img.vec.data(), s->img.scal.data(), path);
}
void sim_get_echo_path(void *h, int16_t *path) {
    SimStream *s = (SimStream *)h;
    GetEchoPath(s->img.vec.data(), s->img.scal.data(), path);
}

// n_blocks consecutive blocks; clean may be NULL.
void sim_process(void *h, const int16_t *far_s, const int16_t *near_s, const int16_t *clean, int16_t *out, int n_blocks) {
    SimStream *s = (SimStream *)h;
    StatePtrs st{s->img.vec.data(), s->img.scal.data(), s->hist.data(), nullptr};
    IoView io{far_s, near_s, clean, out, 0, kBlock};
    if (clean) BlockEngine<SimWave, true>::run_stream(st, io, 0, n_blocks);
    else BlockEngine<SimWave, false>::run_stream(st, io, 0, n_blocks);
}

// The role decomposition of the pipelined kernel's deepest shape (aecm_block_kernels.hip: front, delay, channel, gain and tail
// waves, each with registers of its own, one step apart, meeting through hand-over slots) restated on the simulator: the SAME
// BlockEngine functions, the same state ownership (which role loads and stores what), the same slot rings and the same rules for
// the far-history row -- step by step with every role's reads seeing only what earlier steps wrote.  order = 0: inside a step the
// consumers run first (what a barrier guarantees: nobody sees this step's writes); order = 1: the producers run first -- a role
// that read a slot written in the same step would now see other data, so the two orders (and the plain engine) must agree.
void sim_process_roles(void *h, const int16_t *far_s, const int16_t *near_s, int16_t *out, int n_blocks, int order) {
    using E = BlockEngine<SimWave, false>;
    SimStream *s = (SimStream *)h;
    uint32_t *vec = s->img.vec.data();
    int32_t *scal = s->img.scal.data();
    uint16_t *hist = s->hist.data();
    IoView io{far_s, near_s, nullptr, out, 0, kBlock};
    E::StridedIo sio{io, 0};
    constexpr int kSlots = 4;
    struct Slot { E::Spectrum xf, df; } slots[kSlots];
    int delays[2] = {0, 0};
    VecI far_rows[2];
    E::GainInput gains[2];
    E::TailInput tails[2];
    E::Regs rf, rd, rc, rg, rt;                  // front, delay, channel, gain, tail
    for (E::Regs *r : {&rf, &rd, &rc, &rg, &rt}) E::init_lane_constants(*r, nullptr);
    VecI x_old, d_old, ovl, c_old;
    E::load_time_state(vec, rf.lane, x_old, d_old);
    E::load_delay_state(rd, vec, scal);
    int hist_pos = scal[S_HISTPOS];
    rd.u.fixed_delay = scal[S_FIXED_DELAY];
    E::load_state(rc, vec, scal);
    E::load_state(rg, vec, scal);
    E::load_tail_state(vec, rt.lane, ovl, c_old);
    auto front = [&](int b) {
        if (b < 0 || b >= n_blocks) return;
        const VecI far_new = sio.far(rf, b), near_new = sio.near(rf, b);
        E::Spectrum cf;
        Slot &sl = slots[b % kSlots];
        E::front_block(rf, x_old, far_new, d_old, near_new, VecI(0), VecI(0), sl.xf, sl.df, cf);
        x_old = far_new;
        d_old = near_new;
    };
    auto delay = [&](int b) {
        if (b < 0 || b >= n_blocks) return;
        const Slot &sl = slots[b % kSlots];
        const int estimate = E::delay_block(rd, sl.xf, sl.df);
        delays[b & 1] = estimate;
        hist_pos = hist_pos + 1 >= kHistory ? 0 : hist_pos + 1;
        const int d = E::effective_delay(rd.u, estimate);
        if (d != 0) {
            if (d == 1 && b > 0) far_rows[b & 1] = slots[(b - 1) % kSlots].xf.mag;
            else far_rows[b & 1] = SimWave::load_u16(hist + E::aligned_slot(hist_pos, d) * kLanes, rd.lane);
        }
    };
    auto channel = [&](int b) {
        if (b < 0 || b >= n_blocks) return;
        const Slot &sl = slots[b % kSlots];
        E::update_startup(rc.u);
        E::track_q(rc.u, sl.df, sl.df);
        gains[b & 1] = E::channel_block<true>(rc, hist, sl.xf, sl.df, delays[b & 1], far_rows[b & 1]);
    };
    auto gain = [&](int b) {
        if (b < 0 || b >= n_blocks) return;
        const Slot &sl = slots[b % kSlots];
        E::track_q(rg.u, sl.df, sl.df);
        tails[b & 1] = E::gain_block(rg, sl.df, sl.df, gains[b & 1]);
    };
    auto tail = [&](int b) {
        if (b < 0 || b >= n_blocks) return;
        const E::TailInput &t = tails[b & 1];
        rt.out_ovl = ovl;
        const VecI o = E::tail_block(rt, t.a, t.b, t.clean_q);
        ovl = rt.out_ovl;
        sio.out(rt, b, o);
    };
    for (int step = 0; step < n_blocks + 4; ++step) {             // front: block step, delay: step - 1, channel: - 2, gain: - 3, tail: - 4
        if (order == 0) { tail(step - 4); gain(step - 3); channel(step - 2); delay(step - 1); front(step); }
        else { front(step); delay(step - 1); channel(step - 2); gain(step - 3); tail(step - 4); }
    }
    // the gain wave's part of the state goes to the channel wave, which stores the stream's state; the others store their own
    rc.b.echo_filt = rg.b.echo_filt; rc.b.near_filt = rg.b.near_filt; rc.b.low_ctr = rg.b.low_ctr; rc.b.high_ctr = rg.b.high_ctr;
    rc.b.noise_est = rg.b.noise_est;
    rc.u.seed = rg.u.seed; rc.u.sup_gain = rg.u.sup_gain; rc.u.sup_gain_old = rg.u.sup_gain_old; rc.u.noise_ctr = rg.u.noise_ctr;
    rc.b64.echo_filt = rg.b64.echo_filt; rc.b64.near_filt = rg.b64.near_filt; rc.b64.noise_est = rg.b64.noise_est;
    rc.b64.low_ctr = rg.b64.low_ctr; rc.b64.high_ctr = rg.b64.high_ctr;
    E::store_state<false, false, false>(rc, vec, scal);
    E::store_time_state(vec, rf.lane, x_old, d_old);
    E::store_tail_state(vec, rt.lane, ovl, c_old);
    E::store_delay_state(rd, vec, scal);
}

// The constants blob the HOST builds for the GPU kernels, next to the same quantities evaluated from
// their definitions in aecm_wave.h / the simulator policy (rows: LaneConstRow order, then twiddles).
void sim_constants(uint32_t *blob_out, uint32_t *defined_lane_rows, uint32_t *defined_twiddles) {
    std::vector<uint32_t> blob;
    BuildKernelConstants(&blob);
    memcpy(blob_out, blob.data(), blob.size() * 4);
    BlockEngine<SimWave, false>::Regs r;
    BlockEngine<SimWave, false>::init_lane_constants(r, nullptr);
    for (int k = 0; k < kLaneConstRows; ++k)
        for (int t = 0; t < kLanes; ++t) defined_lane_rows[k * kLanes + t] = (uint32_t)r.lc[k].v[t];
    VecI wre, wim;
    uint32_t *o = defined_twiddles;
    VecI nwre, nwim, sre, cre, sim_, cim;
#define SIM_TW(S) SimWave::inv_twiddles<S>(wre, wim, nwre, nwim); for (int t = 0; t < kLanes; ++t) { *o++ = (uint32_t)wre.v[t]; *o++ = (uint32_t)wim.v[t]; *o++ = (uint32_t)nwre.v[t]; *o++ = (uint32_t)nwim.v[t]; }
    SIM_TW(0) SIM_TW(1) SIM_TW(2) SIM_TW(3) SIM_TW(4) SIM_TW(5) SIM_TW(6)
#undef SIM_TW
#define SIM_FT(S) SimWave::fwd_twiddles<S>(wre, wim, nwre, nwim); for (int t = 0; t < kLanes; ++t) { *o++ = (uint32_t)wre.v[t]; *o++ = (uint32_t)wim.v[t]; *o++ = (uint32_t)nwre.v[t]; *o++ = (uint32_t)nwim.v[t]; }
    SIM_FT(1) SIM_FT(2) SIM_FT(3) SIM_FT(4) SIM_FT(5) SIM_FT(6)
#undef SIM_FT
#define SIM_FO(S) SimWave::fwd_offsets<S>(sre, cre, sim_, cim); for (int t = 0; t < kLanes; ++t) { *o++ = (uint32_t)sre.v[t]; *o++ = (uint32_t)cre.v[t]; *o++ = (uint32_t)sim_.v[t]; *o++ = (uint32_t)cim.v[t]; }
    SIM_FO(2) SIM_FO(4) SIM_FO(6)
#undef SIM_FO
}

// One 128-point transform of the kernel's fft128 on natural-order data (re/im in, re/im out): lane t
// starts with points t and t + 64 and ends with bins bitrev6(t) and bitrev6(t) + 64.  variant:
// 0 = forward of real input (im ignored), 1 = forward complex, 2 = inverse.  The last stage only
// produces what its caller consumes: forward -> im of bins >= 64 is not computed (returned as 0),
// inverse -> real parts only.  Returns the inverse transform's accumulated scale (0 for forward).
int sim_fft128(int16_t *re, int16_t *im, int variant) {
    using E = BlockEngine<SimWave, false>;
    VecI a, b;
    for (int t = 0; t < kLanes; ++t) {
        const int im_a = variant == 0 ? 0 : im[t], im_b = variant == 0 ? 0 : im[t + 64];
        a.v[t] = (re[t] & 0xffff) | (int)((unsigned)im_a << 16);
        b.v[t] = (re[t + 64] & 0xffff) | (int)((unsigned)im_b << 16);
    }
    int scale = 0;
    const VecI kp = SimWave::opaque_const(32770);
    if (variant == 0) scale = E::fft128<false, true>(a, b, kp);
    else if (variant == 1) scale = E::fft128<false, false>(a, b, kp);
    else {
        scale = E::fft128<true, false>(a, b, kp);
        for (int t = 0; t < kLanes; ++t) {          // the inverse transform hands its real parts on in the upper halves
            a.v[t] >>= 16;
            b.v[t] >>= 16;
        }
    }
    for (int t = 0; t < kLanes; ++t) {
        int r = 0;
        for (int k = 0; k < 6; ++k) r |= ((t >> k) & 1) << (5 - k);
        re[r] = (int16_t)a.v[t];
        re[r + 64] = (int16_t)b.v[t];
        im[r] = variant == 2 ? (int16_t)0 : (int16_t)(a.v[t] >> 16);
        im[r + 64] = 0;
    }
    return scale;
}

void sim_digest(void *h, uint32_t *digest) {
    SimStream *s = (SimStream *)h;
    ComputeDigest(s->img.vec.data(), s->img.scal.data(), s->hist.data(), digest);
}

}  // extern "C"
