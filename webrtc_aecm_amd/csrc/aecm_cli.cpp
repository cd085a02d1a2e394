// aecm_run -- command-line front end of the MI355X AECM engine.
//
//   aecm_run far.wav near.wav            one pair through the drop-in session ABI, exactly the
//                                        reference CLI's procedure (main.cc:97-191): cng on, echoMode 1,
//                                        msInSndCardBuf 40, min(160, rate/100)-sample calls, mono 16-bit,
//                                        result in <near>_out<ext>, tail beyond the last full call untouched
//   aecm_run --batch list.txt            many pairs at once (one "far.wav near.wav" per line): all
//                                        recordings of one sample rate are processed as one device batch
//                                        (WebRtcAecmBatch_ProcessRecordingsHost), one wavefront per file
//   aecm_run --decode in.wav out.wav     read in.wav the way the pair modes do and write it back as 16-bit PCM
//                                        (which sample formats are read, and how they become int16: ReadWav)
// RIFF/WAVE reader for the sample formats the reference CLI accepts, 16-bit PCM writer (the reference uses dr_wav.h for
// I/O only).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/aecm_batch.h"
#include "../../include/echo_control_mobile.h"

namespace {

struct Wav {
    uint32_t rate = 0;
    uint16_t channels = 0;
    std::vector<int16_t> samples;
};

// Sample formats the reference CLI accepts through dr_wav's drwav_open_file_and_read_pcm_frames_s16 (main.cc:39-54;
// dr_wav.h v0.12 is vendored in the reference for I/O only) and how each becomes int16 -- restated from its published
// conversion rules, because a different rounding of a float or 24-bit input would be a different input to the canceller:
//   PCM  8 bit  unsigned: (x << 8) - 32768                         (dr_wav.h: drwav_u8_to_s16)
//   PCM 16 bit            as it is
//   PCM 24 / 32 bit       the upper 16 bits (arithmetic shift)      (drwav_s24_to_s16, drwav_s32_to_s16)
//   PCM 40..64 bit        the upper 16 bits                         (drwav__pcm_to_s16, generic path)
//   IEEE float 32 / 64    c = clamp(x, -1, 1) + 1;  (int)(c * 32767.5) - 32768, evaluated in the sample's own precision
//   A-law, mu-law         ITU-T G.711 expansion (the tables of dr_wav.h are that expansion; generated here)
// RIFF containers, plain and WAVE_FORMAT_EXTENSIBLE headers.  Not read: ADPCM variants, Sony Wave64, RF64.
enum : uint16_t { kFmtPcm = 1, kFmtAdpcm = 2, kFmtFloat = 3, kFmtAlaw = 6, kFmtMulaw = 7, kFmtExtensible = 0xFFFE };

int16_t AlawToS16(uint8_t byte) {
    const unsigned a = byte ^ 0x55u, exponent = (a >> 4) & 7u, mantissa = a & 15u;
    const int magnitude = exponent == 0 ? (int)(mantissa << 4) + 8 : (int)(((mantissa << 4) + 0x108u) << (exponent - 1));
    return (int16_t)((a & 0x80u) ? magnitude : -magnitude);
}

int16_t MulawToS16(uint8_t byte) {
    const unsigned u = (uint8_t)~byte, exponent = (u >> 4) & 7u, mantissa = u & 15u;
    const int magnitude = (int)((((mantissa << 3) + 0x84u) << exponent) - 0x84u);
    return (int16_t)((u & 0x80u) ? -magnitude : magnitude);
}

// One sample of `bytes` bytes (little endian) of format `format` -> int16.  False: a format / width dr_wav does not convert.
bool DecodeSample(uint16_t format, unsigned bytes, const uint8_t *p, int16_t *out) {
    switch (format) {
        case kFmtPcm: {
            if (bytes == 1) { *out = (int16_t)(((int)p[0] << 8) - 32768); return true; }
            if (bytes < 2 || bytes > 8) return false;
            *out = (int16_t)(uint16_t)(p[bytes - 2] | (unsigned)p[bytes - 1] << 8);       // the upper 16 bits of the sample
            return true;
        }
        case kFmtFloat: {
            if (bytes == 4) {
                float x;
                memcpy(&x, p, 4);
                volatile float c = (x < -1.0f) ? -1.0f : ((x > 1.0f) ? 1.0f : x);   // volatile: every step rounded to float, no fusing
                c = c + 1.0f;
                c = c * 32767.5f;
                *out = (int16_t)((int)c - 32768);
                return true;
            }
            if (bytes == 8) {
                double x;
                memcpy(&x, p, 8);
                volatile double c = (x < -1.0) ? -1.0 : ((x > 1.0) ? 1.0 : x);
                c = c + 1.0;
                c = c * 32767.5;
                *out = (int16_t)((int)c - 32768);
                return true;
            }
            return false;
        }
        case kFmtAlaw: if (bytes != 1) return false; *out = AlawToS16(p[0]); return true;
        case kFmtMulaw: if (bytes != 1) return false; *out = MulawToS16(p[0]); return true;
        default: return false;
    }
}

bool ReadWav(const std::string &path, Wav *w) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    auto rd32 = [&](uint32_t *v) { return fread(v, 4, 1, f) == 1; };
    char tag[4];
    uint32_t riff_size = 0;
    bool ok = fread(tag, 1, 4, f) == 4 && memcmp(tag, "RIFF", 4) == 0 && rd32(&riff_size) && fread(tag, 1, 4, f) == 4 &&
              memcmp(tag, "WAVE", 4) == 0;
    bool have_fmt = false, have_data = false;
    uint16_t format = 0, bits = 0, block_align = 0;
    while (ok && !have_data) {
        uint32_t size = 0;
        if (fread(tag, 1, 4, f) != 4 || !rd32(&size)) break;
        if (memcmp(tag, "fmt ", 4) == 0 && size >= 16) {
            uint8_t b[16];
            ok = fread(b, 1, 16, f) == 16;
            format = (uint16_t)(b[0] | b[1] << 8);
            w->channels = (uint16_t)(b[2] | b[3] << 8);
            w->rate = (uint32_t)b[4] | (uint32_t)b[5] << 8 | (uint32_t)b[6] << 16 | (uint32_t)b[7] << 24;
            block_align = (uint16_t)(b[12] | b[13] << 8);
            bits = (uint16_t)(b[14] | b[15] << 8);
            if (format == kFmtExtensible && size >= 26) {      // WAVE_FORMAT_EXTENSIBLE: the sub-format GUID starts with the tag
                uint8_t ext[10];
                ok = ok && fread(ext, 1, 10, f) == 10;
                format = (uint16_t)(ext[8] | ext[9] << 8);
                fseek(f, (long)(size - 26 + (size & 1)), SEEK_CUR);
            } else {
                fseek(f, (long)(size - 16 + (size & 1)), SEEK_CUR);
            }
            have_fmt = true;
        } else if (memcmp(tag, "data", 4) == 0) {
            // bytes per sample as the reference CLI's reader takes them (dr_wav.h:1815-1827, drwav_get_bytes_per_pcm_frame): from
            // the bit depth when that is a whole number of bytes, else from the block alignment
            if (!have_fmt || w->channels == 0) { ok = false; break; }
            unsigned bytes = 0;
            if (bits != 0 && bits % 8 == 0) bytes = bits / 8;
            else if (block_align && block_align % w->channels == 0) bytes = block_align / w->channels;
            int16_t probe;
            const uint8_t zero[8] = {0};
            if (bytes == 0 || bytes > 8 || !DecodeSample(format, bytes, zero, &probe)) { ok = false; break; }
            if (format == kFmtPcm && bytes == 2) {              // the usual case: straight into place
                w->samples.resize(size / 2);
                w->samples.resize(fread(w->samples.data(), 2, w->samples.size(), f));
            } else {
                std::vector<uint8_t> raw(size);
                raw.resize(fread(raw.data(), 1, raw.size(), f));
                w->samples.resize(raw.size() / bytes);
                for (size_t i = 0; i < w->samples.size(); ++i) DecodeSample(format, bytes, raw.data() + i * bytes, &w->samples[i]);
            }
            have_data = true;
        } else {
            fseek(f, (long)(size + (size & 1)), SEEK_CUR);
        }
    }
    fclose(f);
    return ok && have_fmt && have_data;
}

bool WriteWav(const std::string &path, uint32_t rate, const std::vector<int16_t> &s) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const uint32_t data_bytes = (uint32_t)(s.size() * 2), riff = 36 + data_bytes, fmt_size = 16, byte_rate = rate * 2;
    const uint16_t pcm = 1, ch = 1, align = 2, bits = 16;
    bool ok = fwrite("RIFF", 1, 4, f) == 4 && fwrite(&riff, 4, 1, f) == 1 && fwrite("WAVEfmt ", 1, 8, f) == 8 &&
              fwrite(&fmt_size, 4, 1, f) == 1 && fwrite(&pcm, 2, 1, f) == 1 && fwrite(&ch, 2, 1, f) == 1 &&
              fwrite(&rate, 4, 1, f) == 1 && fwrite(&byte_rate, 4, 1, f) == 1 && fwrite(&align, 2, 1, f) == 1 &&
              fwrite(&bits, 2, 1, f) == 1 && fwrite("data", 1, 4, f) == 4 && fwrite(&data_bytes, 4, 1, f) == 1 &&
              fwrite(s.data(), 2, s.size(), f) == s.size();
    fclose(f);
    return ok;
}

// "<dir><name>_out<ext>" next to the near-end file (main.cc:186-190).
std::string OutName(const std::string &near_path) {
    const size_t slash = near_path.find_last_of("/\\");
    const size_t dot = near_path.find_last_of('.');
    if (dot == std::string::npos || (slash != std::string::npos && dot < slash)) return near_path + "_out";
    return near_path.substr(0, dot) + "_out" + near_path.substr(dot);
}

constexpr int16_t kEchoMode = 1, kMsInSndCardBuf = 40;       // main.cc:163-164

// main.cc:97-147 through the session ABI.
int ProcessPair(Wav &far_w, Wav &near_w) {
    const size_t samples = std::min<size_t>(160, near_w.rate / 100);
    if (samples == 0) return -1;
    const size_t n_calls = near_w.samples.size() / samples;
    void *inst = WebRtcAecm_Create();
    if (!inst) { fprintf(stderr, "WebRtcAecm_Create failed (no usable GPU?)\n"); return -1; }
    if (WebRtcAecm_Init(inst, (int32_t)near_w.rate) != 0) { printf("WebRtcAecm_Init fail\n"); WebRtcAecm_Free(inst); return -1; }
    AecmConfig cfg;
    cfg.cngMode = AecmTrue;
    cfg.echoMode = kEchoMode;
    if (WebRtcAecm_set_config(inst, cfg) != 0) { printf("WebRtcAecm_set_config fail\n"); WebRtcAecm_Free(inst); return -1; }
    int16_t out[160];
    if (far_w.samples.size() < n_calls * samples) far_w.samples.resize(n_calls * samples, 0);
    for (size_t i = 0; i < n_calls; ++i) {
        int16_t *nearp = near_w.samples.data() + i * samples;
        if (WebRtcAecm_BufferFarend(inst, far_w.samples.data() + i * samples, samples) != 0) {
            printf("WebRtcAecm_BufferFarend() failed.");
            WebRtcAecm_Free(inst);
            return -1;
        }
        if (WebRtcAecm_Process(inst, nearp, nullptr, out, samples, kMsInSndCardBuf) != 0) {
            printf("failed in WebRtcAecm_Process\n");
            WebRtcAecm_Free(inst);
            return -1;
        }
        memcpy(nearp, out, samples * sizeof(int16_t));
    }
    WebRtcAecm_Free(inst);
    return 1;
}

int RunBatch(const char *list_path, const std::vector<int> &devices) {
    FILE *f = fopen(list_path, "r");
    if (!f) { fprintf(stderr, "cannot open %s\n", list_path); return 1; }
    struct Job { std::string far_path, near_path; Wav far_w, near_w; };
    std::vector<Job> jobs;
    char a[2048], b[2048];
    while (fscanf(f, "%2047s %2047s", a, b) == 2) {
        Job j;
        j.far_path = a;
        j.near_path = b;
        if (!ReadWav(j.far_path, &j.far_w) || !ReadWav(j.near_path, &j.near_w) || j.far_w.channels != 1 ||
            j.near_w.channels != 1 || (j.near_w.rate != 8000 && j.near_w.rate != 16000)) {
            fprintf(stderr, "skipping %s %s (need mono 16-bit PCM at 8 or 16 kHz)\n", a, b);
            continue;
        }
        jobs.push_back(std::move(j));
    }
    fclose(f);
    const auto t0 = std::chrono::steady_clock::now();
    std::map<uint32_t, std::vector<size_t>> by_rate;
    for (size_t i = 0; i < jobs.size(); ++i) by_rate[jobs[i].near_w.rate].push_back(i);
    for (auto &kv : by_rate) {
        const uint32_t rate = kv.first;
        const std::vector<size_t> &ids = kv.second;
        const int frame = (int)std::min<uint32_t>(160, rate / 100);
        size_t max_calls = 0;
        for (size_t id : ids) max_calls = std::max(max_calls, jobs[id].near_w.samples.size() / frame);
        const size_t stride = max_calls * frame;
        if (stride == 0) continue;
        std::vector<int16_t> far_all(ids.size() * stride, 0), near_all(ids.size() * stride, 0), out_all(ids.size() * stride, 0);
        for (size_t k = 0; k < ids.size(); ++k) {
            const Job &j = jobs[ids[k]];
            const size_t n = (j.near_w.samples.size() / frame) * frame;
            memcpy(&near_all[k * stride], j.near_w.samples.data(), n * 2);
            memcpy(&far_all[k * stride], j.far_w.samples.data(), std::min(n, j.far_w.samples.size()) * 2);
        }
        // Recordings are independent: static contiguous shards, one host thread and one AecmBatch per device, no
        // exchange between devices (the C++ form of the multi-GPU sharding of DESIGN.md section 6).
        const size_t n_dev = devices.size();
        std::vector<int32_t> shard_rc(n_dev, 0);
        std::vector<double> shard_s(n_dev, 0.0);              // per device: seconds in the library, recordings, WebRtcAecm_ProcessBlock-equivalents
        std::vector<size_t> shard_n(n_dev, 0);
        std::vector<std::thread> workers;
        for (size_t d = 0; d < n_dev; ++d) {
            const size_t base = ids.size() / n_dev, rem = ids.size() % n_dev;
            const size_t first = d * base + std::min(d, rem), count = base + (d < rem ? 1 : 0);
            if (count == 0) continue;
            workers.emplace_back([&, d, first, count] {
                AecmBatch *batch = WebRtcAecmBatch_Create((int32_t)count, devices[d]);
                if (!batch) { shard_rc[d] = -1; return; }
                AecmConfig cfg;
                cfg.cngMode = AecmTrue;
                cfg.echoMode = kEchoMode;
                const auto s0 = std::chrono::steady_clock::now();
                int32_t rc = WebRtcAecmBatch_Init(batch, (int32_t)rate);
                if (rc == 0) rc = WebRtcAecmBatch_set_config(batch, cfg, 0, -1);
                if (rc == 0)
                    rc = WebRtcAecmBatch_ProcessRecordingsHost(batch, &far_all[first * stride], &near_all[first * stride], /*nearendClean*/ nullptr,
                                                               &out_all[first * stride], (int64_t)stride, frame, (int32_t)max_calls,
                                                               kMsInSndCardBuf);
                shard_s[d] = std::chrono::duration<double>(std::chrono::steady_clock::now() - s0).count();
                shard_n[d] = count;
                WebRtcAecmBatch_Free(batch);
                shard_rc[d] = rc;
            });
        }
        for (std::thread &w : workers) w.join();
        for (size_t d = 0; d < n_dev; ++d) {
            if (shard_rc[d] == -1) { fprintf(stderr, "WebRtcAecmBatch_Create failed on device %d (no usable GPU?)\n", devices[d]); return 1; }
            if (shard_rc[d] != 0) { fprintf(stderr, "batch at %u Hz failed on device %d: %d\n", rate, devices[d], shard_rc[d]); return 1; }
        }
        // the counters of the shards, gathered where the threads join (no collective needed inside one process)
        for (size_t d = 0; d < n_dev; ++d)
            if (shard_n[d] != 0)
                printf("device %d: %zu recordings at %u Hz, %zu frames of 64 samples each, %.3f s in the library (%.2f M frames/s)\n", devices[d], shard_n[d], rate,
                       stride / 64, shard_s[d], shard_s[d] > 0 ? (double)shard_n[d] * (double)(stride / 64) / shard_s[d] / 1e6 : 0.0);
        for (size_t k = 0; k < ids.size(); ++k) {
            Job &j = jobs[ids[k]];
            const size_t n = (j.near_w.samples.size() / frame) * frame;      // tail stays untouched (main.cc:111)
            memcpy(j.near_w.samples.data(), &out_all[k * stride], n * 2);
        }
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("time interval: %d ms (%zu recordings on %zu device shard(s))\n", (int)ms, jobs.size(), devices.size());
    for (Job &j : jobs)
        if (!WriteWav(OutName(j.near_path), j.near_w.rate, j.near_w.samples)) { fprintf(stderr, "ERROR\n"); return 1; }
    return 0;
}

}  // namespace

int main(int argc, char **argv) {
    printf("WebRTC Acoustic Echo Canceller for Mobile -- MI355X engine\n");
    printf("usage : aecm_run [--device N] far_file.wav near_file.wav | aecm_run --batch pairs.txt [--devices 0,1,...] | aecm_run --decode in.wav out.wav\n");
    if (argc >= 3 && strcmp(argv[1], "--device") == 0) {     // the HIP device of the single-pair form (WebRtcAecm_Create has no device argument)
        WebRtcAecm_SetDefaultDevice(atoi(argv[2]));
        argv += 2;
        argc -= 2;
    }
    if (argc < 3) return -1;
    if (strcmp(argv[1], "--decode") == 0) {                  // the reader on its own (no engine, no GPU)
        Wav w;
        if (argc < 4 || !ReadWav(argv[2], &w)) { printf("failed to read wav files.\n"); return 1; }
        if (w.channels != 1) { printf("mono files only.\n"); return 1; }
        return WriteWav(argv[3], w.rate, w.samples) ? 0 : 1;
    }
    if (strcmp(argv[1], "--batch") == 0) {
        // --devices: HIP device ids to shard the recordings over (one host thread + one batch each); default: device 0
        std::vector<int> devices;
        if (argc >= 5 && strcmp(argv[3], "--devices") == 0) {
            for (const char *p = argv[4]; *p;) {
                devices.push_back(atoi(p));
                while (*p && *p != ',') ++p;
                if (*p == ',') ++p;
            }
        }
        if (devices.empty()) devices.push_back(0);
        return RunBatch(argv[2], devices);
    }
    Wav far_w, near_w;
    if (!ReadWav(argv[2], &near_w) || !ReadWav(argv[1], &far_w)) { printf("failed to read wav files.\n"); return 1; }
    if (near_w.channels != 1 || far_w.channels != 1) { printf("mono files only.\n"); return 1; }   // main.cc:47-52
    const auto t0 = std::chrono::steady_clock::now();
    ProcessPair(far_w, near_w);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("time interval: %d ms\n ", (int)ms);                                                     // main.cc:168
    if (!WriteWav(OutName(argv[2]), near_w.rate, near_w.samples)) { fprintf(stderr, "ERROR\n"); return 1; }
    return 0;
}
