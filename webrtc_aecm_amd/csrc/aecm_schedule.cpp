// Index-domain run of the session machinery: which far/near samples feed which block, and where
// each output sample comes from (aecm_session_flow.h).
#include "aecm_session_flow.h"

namespace aecm {

RecordingSchedule BuildRecordingSchedule(int fs, int frame, int n_calls, int16_t ms) {
    RecordingSchedule s;
    SessionFlow<int32_t> flow(/*zero=*/-1);
    s.first_error = flow.Init(fs);
    if (s.first_error) return s;
    s.out_map.resize((size_t)n_calls * frame);
    std::vector<int32_t> tags(frame), out(frame);
    for (int c = 0; c < n_calls; ++c) {
        for (int i = 0; i < frame; ++i) tags[i] = c * frame + i;       // far and near carry the same index space
        int32_t rc = flow.BufferFarend(tags.data(), (size_t)frame);
        bool passthrough = false;
        if (rc == 0) {
            rc = flow.Process(tags.data(), nullptr, out.data(), (size_t)frame, ms,
                              [&](const int32_t *far_b, const int32_t *near_b, const int32_t *, int32_t *out_b, int n) {
                                  s.far_map.insert(s.far_map.end(), far_b, far_b + n * kBlock);
                                  s.near_map.insert(s.near_map.end(), near_b, near_b + n * kBlock);
                                  for (int k = 0; k < n * kBlock; ++k) out_b[k] = s.n_blocks * kBlock + k;
                                  s.n_blocks += n;
                                  return true;
                              },
                              &passthrough);
        }
        if (rc == kWarnBadParameter) s.warned = true;
        else if (rc != 0 && s.first_error == 0) s.first_error = rc;
        for (int i = 0; i < frame; ++i) {
            const int32_t v = out[i];
            // start-up calls copy the near input through: re-tag into the "near sample" range
            s.out_map[(size_t)c * frame + i] = passthrough ? -(v + 2) : v;
        }
    }
    return s;
}

}  // namespace aecm
