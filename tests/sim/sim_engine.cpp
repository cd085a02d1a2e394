// TEST INFRASTRUCTURE -- an implementation of aecm::BatchEngine (webrtc_aecm_amd/csrc/aecm_engine.h)
// on the 64-lane CPU simulator, linked ONLY into tests/_build/libaecm_sim.so so that the product's
// host-side session logic (aecm_session.cpp: jitter buffer, start-up gating, 80->64 re-blocking) can
// be exercised by the CPU-only tests.  The shipped library links aecm_engine.cpp (HIP) instead.
#include <string.h>

#include <vector>

#include "wave_sim.h"
#include "aecm_engine.h"
#include "aecm_session.h"
#include "aecm_session_flow.h"
#include "aecm_wave.h"

namespace aecm {

namespace {
struct SimStore {
    std::vector<uint32_t> vec;
    std::vector<int32_t> scal;
    std::vector<uint16_t> hist;
};
SimStore *Store(const StatePtrs &st) { return reinterpret_cast<SimStore *>(st.vec); }
}  // namespace

BatchEngine *BatchEngine::Create(int num_streams, int) {
    BatchEngine *e = new BatchEngine();
    e->num_streams_ = num_streams;
    SimStore *s = new SimStore();
    s->vec.resize((size_t)num_streams * kVecWordsPerStream);
    s->scal.resize((size_t)num_streams * kNumScal);
    s->hist.resize((size_t)num_streams * kHistWordsPerStream);
    e->st_.vec = reinterpret_cast<uint32_t *>(s);
    return e;
}
BatchEngine::~BatchEngine() { delete Store(st_); }

bool BatchEngine::Init(int fs) {
    StreamImage img;
    if (!BuildInitImage(fs, &img)) return false;
    SimStore *s = Store(st_);
    for (int i = 0; i < num_streams_; ++i) {
        memcpy(&s->vec[(size_t)i * kVecWordsPerStream], img.vec.data(), kVecWordsPerStream * sizeof(uint32_t));
        memcpy(&s->scal[(size_t)i * kNumScal], img.scal.data(), kNumScal * sizeof(int32_t));
    }
    std::fill(s->hist.begin(), s->hist.end(), 0);
    initialized_ = true;
    fs_ = fs;
    return true;
}
bool BatchEngine::PatchScalars(const int32_t *f, const int32_t *v, int n, int first, int count) {
    if (count < 0) count = num_streams_ - first;
    SimStore *s = Store(st_);
    for (int i = first; i < first + count; ++i)
        for (int k = 0; k < n; ++k) s->scal[(size_t)i * kNumScal + f[k]] = v[k];
    return true;
}
bool BatchEngine::SetConfig(int cng, int em, int first, int count) {
    int32_t scal[kNumScal] = {0};
    if (!ApplyConfig(scal, cng, em)) return false;
    const int32_t fields[7] = {S_CNG, S_SUPGAIN, S_SUPGAIN_OLD, S_SG_A, S_SG_D, S_SG_DAB, S_SG_DBD};
    int32_t values[7];
    for (int i = 0; i < 7; ++i) values[i] = scal[fields[i]];
    return PatchScalars(fields, values, 7, first, count);
}
bool BatchEngine::SetCngMode(int cng, int first, int count) {
    const int32_t f[1] = {S_CNG}, v[1] = {cng};
    return PatchScalars(f, v, 1, first, count);
}
bool BatchEngine::Control(int fixed_delay, int nlp, int first, int count) {
    int32_t scal[kNumScal] = {0};
    ApplyControl(scal, fixed_delay, nlp);
    const int32_t f[2] = {S_NLP, S_FIXED_DELAY}, v[2] = {scal[S_NLP], scal[S_FIXED_DELAY]};
    return PatchScalars(f, v, 2, first, count);
}
bool BatchEngine::ProcessBlocksHost(const IoView &io, int num_blocks) {
    SimStore *s = Store(st_);
    StatePtrs st{s->vec.data(), s->scal.data(), s->hist.data(), nullptr};
    for (int i = 0; i < num_streams_; ++i) {
        if (io.near_clean) BlockEngine<SimWave, true>::run_stream(st, io, i, num_blocks);
        else BlockEngine<SimWave, false>::run_stream(st, io, i, num_blocks);
    }
    return true;
}
bool BatchEngine::ProcessBlocks(const IoView &io, int num_blocks, const int32_t *) { return ProcessBlocksHost(io, num_blocks); }
bool BatchEngine::Synchronize() { return true; }
bool BatchEngine::HarvestTimers(bool) { return true; }
bool BatchEngine::LastLaunchMs(float *ms) { *ms = 0.f; return true; }
bool BatchEngine::Timers(double *t, int64_t *n) { *t = 0; *n = 0; return true; }
void BatchEngine::ResetTimers() {}
bool BatchEngine::SetEchoPath(int stream, const int16_t path[kBins]) {
    SimStore *s = Store(st_);
    aecm::SetEchoPath(&s->vec[(size_t)stream * kVecWordsPerStream], &s->scal[(size_t)stream * kNumScal], path);
    return true;
}
bool BatchEngine::GetEchoPath(int stream, int16_t path[kBins]) {
    SimStore *s = Store(st_);
    aecm::GetEchoPath(&s->vec[(size_t)stream * kVecWordsPerStream], &s->scal[(size_t)stream * kNumScal], path);
    return true;
}
bool BatchEngine::ExportState(int, void *) { return false; }
int32_t BatchEngine::ImportState(int, const void *) { return kErrUnspecified; }
bool BatchEngine::Digest(int stream, uint32_t d[kDigestWords]) {
    SimStore *s = Store(st_);
    ComputeDigest(&s->vec[(size_t)stream * kVecWordsPerStream], &s->scal[(size_t)stream * kNumScal],
                  &s->hist[(size_t)stream * kHistWordsPerStream], d);
    return true;
}

}  // namespace aecm

namespace aecm {
// Same algorithm as BatchEngine::ProcessRecordings in aecm_engine.cpp, host loops instead of kernels.
bool BatchEngine::ProcessRecordings(const int16_t *far, const int16_t *near, const int16_t *clean, int16_t *out, int64_t stride,
                                    int frame, int n_calls, int16_t ms, bool, int32_t *rc) {
    const RecordingSchedule sch = BuildRecordingSchedule(fs_, frame, n_calls, ms);
    if (sch.first_error) { *rc = sch.first_error; return true; }
    *rc = sch.warned ? kWarnBadParameter : 0;
    const int64_t n_blk = (int64_t)sch.n_blocks * kBlock, n_in = (int64_t)n_calls * frame;
    std::vector<int16_t> bfar(num_streams_ * n_blk), bnear(num_streams_ * n_blk), bclean(clean ? num_streams_ * n_blk : 0),
        bout(num_streams_ * n_blk);
    for (int s = 0; s < num_streams_; ++s)
        for (int64_t j = 0; j < n_blk; ++j) {
            bfar[s * n_blk + j] = sch.far_map[j] >= 0 ? far[s * stride + sch.far_map[j]] : 0;
            bnear[s * n_blk + j] = sch.near_map[j] >= 0 ? near[s * stride + sch.near_map[j]] : 0;
            if (clean) bclean[s * n_blk + j] = sch.near_map[j] >= 0 ? clean[s * stride + sch.near_map[j]] : 0;
        }
    if (sch.n_blocks > 0) {
        IoView io{bfar.data(), bnear.data(), clean ? bclean.data() : nullptr, bout.data(), n_blk, kBlock};
        ProcessBlocksHost(io, sch.n_blocks);
    }
    const int16_t *pass = clean ? clean : near;      // pass-through source (echo_control_mobile.cc:285-291)
    for (int s = 0; s < num_streams_; ++s)
        for (int64_t j = 0; j < n_in; ++j) {
            const int32_t v = sch.out_map[j];
            out[s * stride + j] = v >= 0 ? bout[s * n_blk + v] : (v == -1 ? (int16_t)0 : pass[s * stride + (-(int64_t)v - 2)]);
        }
    return true;
}
}  // namespace aecm

extern "C" int32_t sim_recordings(int n_streams, int n_samples, int fs, int frame, int cng, int echo_mode, int ms,
                                  const int16_t *far, const int16_t *near, const int16_t *clean, int16_t *out) {
    aecm::BatchEngine *e = aecm::BatchEngine::Create(n_streams, 0);
    e->Init(fs);
    e->SetConfig(cng, echo_mode, 0, n_streams);
    int32_t rc = -1;
    e->ProcessRecordings(far, near, clean, out, n_samples, frame, n_samples / frame, (int16_t)ms, true, &rc);
    delete e;
    return rc;
}

// C doors onto the product's Session class for the CPU tests (same shapes as the public ABI).
extern "C" {
void *simsession_create() { return aecm::Session::Create(); }
void simsession_free(void *h) { delete static_cast<aecm::Session *>(h); }
int32_t simsession_init(void *h, int32_t fs) { return static_cast<aecm::Session *>(h)->Init(fs); }
int32_t simsession_buffer_farend(void *h, const int16_t *f, size_t n) { return static_cast<aecm::Session *>(h)->BufferFarend(f, n); }
int32_t simsession_process(void *h, const int16_t *d, const int16_t *c, int16_t *out, size_t n, int16_t ms) {
    return static_cast<aecm::Session *>(h)->Process(d, c, out, n, ms);
}
int32_t simsession_set_config(void *h, int16_t cng, int16_t em) { return static_cast<aecm::Session *>(h)->SetConfig(cng, em); }
int32_t simsession_init_echo_path(void *h, const void *p, size_t n) { return static_cast<aecm::Session *>(h)->InitEchoPath(p, n); }
int32_t simsession_get_echo_path(void *h, void *p, size_t n) { return static_cast<aecm::Session *>(h)->GetEchoPath(p, n); }
}
