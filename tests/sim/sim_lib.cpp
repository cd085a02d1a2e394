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
    StatePtrs st{s->img.vec.data(), s->img.scal.data(), s->hist.data()};
    IoView io{far_s, near_s, clean, out, 0, kBlock};
    if (clean) BlockEngine<SimWave, true>::run_stream(st, io, 0, n_blocks);
    else BlockEngine<SimWave, false>::run_stream(st, io, 0, n_blocks);
}

void sim_digest(void *h, uint32_t *digest) {
    SimStream *s = (SimStream *)h;
    ComputeDigest(s->img.vec.data(), s->img.scal.data(), s->hist.data(), digest);
}

}  // extern "C"
