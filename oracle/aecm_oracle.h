/*
 * TEST INFRASTRUCTURE -- NOT PART OF THE SHIPPED PRODUCT.
 *
 * CPU oracle: a plain-C restatement of the reference's AECM block path
 * (WebRtcAecm_ProcessBlock and everything it calls, reference aecm/aecm_core_c.cc:368-711).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only as
 * the checker.  The product (webrtc_aecm_amd/) never includes, links or calls anything here.
 *
 * Parity pinning: this restatement is checked bit-for-bit against the UNMODIFIED reference
 * compiled from /root/reference (oracle/_ref/libaecm_ref.so, recipe in oracle/Makefile) and
 * against the committed golden vectors under tests/golden/ (generated from that same build by
 * tools/gen_golden.py).  See tests/test_oracle_*.py.
 */
#ifndef AECM_ORACLE_H_
#define AECM_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    ORC_BLOCK = 64,      /* PART_LEN   (reference aecm/aecm_defines.h:19) */
    ORC_BINS = 65,       /* PART_LEN1  (aecm_defines.h:22) */
    ORC_HISTORY = 100,   /* MAX_DELAY  (aecm_defines.h:26) */
    ORC_LOGBUF = 64,     /* MAX_BUF_LEN (aecm_defines.h:33) */
    ORC_DIGEST_WORDS = 24
};
/* Branch statistics of the blocks processed since aecm_oracle_init (bench.py's content sweep): how many blocks ... */
enum {
    ORC_STAT_BLOCKS = 0,
    ORC_STAT_NLMS,            /* ... ran the NLMS channel update (mu != 0, aecm_core.cc:830) */
    ORC_STAT_GAIN_ZERO,       /* ... had supGain == 0 after CalcSuppressionGain (every Wiener gain is then ONE_Q14) */
    ORC_STAT_Q_STEADY,        /* ... did not raise the near-end Q domain (dfaCleanQDomain <= dfaCleanQDomainOld, aecm_core_c.cc:552-579) */
    ORC_STAT_IFFT_UNSCALED,   /* ... went through the inverse transform without a rescaling stage (complex_fft.c:382-396) */
    ORC_STAT_DELAYED,         /* ... used an aligned far spectrum older than the current one (delay != 0, aecm_core.cc:157-172) */
    ORC_STAT_VAD,             /* ... had currentVADValue set (aecm_core.cc:733-740) */
    ORC_STATS = 8
};

typedef struct AecmOracle AecmOracle;

AecmOracle *aecm_oracle_create(void);
void aecm_oracle_free(AecmOracle *o);

/* WebRtcAecm_InitCore (aecm_core.cc:358-473) followed by the wrapper's default configuration
 * set_config(cng=1, echoMode=3) (echo_control_mobile.cc:183-188).  fs must be 8000 or 16000;
 * returns 0, or -1 for any other rate. */
int aecm_oracle_init(AecmOracle *o, int fs);

/* The part of WebRtcAecm_set_config that lands in the core (echo_control_mobile.cc:424-476).
 * Returns 0, or -1 on a bad parameter. */
int aecm_oracle_set_config(AecmOracle *o, int cng_mode, int echo_mode);

/* WebRtcAecm_Control (aecm_core.cc:477-482): never called by the reference's own callers. */
void aecm_oracle_control(AecmOracle *o, int fixed_delay, int nlp_flag);

/* WebRtcAecm_InitEchoPathCore (aecm_core.cc:249-265) / read-back of channelStored. */
void aecm_oracle_init_echo_path(AecmOracle *o, const int16_t path[ORC_BINS]);
void aecm_oracle_get_echo_path(const AecmOracle *o, int16_t path[ORC_BINS]);

/* WebRtcAecm_ProcessBlock (aecm_core_c.cc:368-711).  near_clean may be NULL.  Returns 0. */
int aecm_oracle_process_block(AecmOracle *o,
                              const int16_t far_blk[ORC_BLOCK],
                              const int16_t near_noisy[ORC_BLOCK],
                              const int16_t *near_clean,
                              int16_t out[ORC_BLOCK]);

/* Convenience: run n_blocks consecutive blocks of one stream (contiguous samples). */
int aecm_oracle_process_stream(AecmOracle *o, const int16_t *far_s, const int16_t *near_s,
                               int16_t *out, size_t n_blocks);

/* State digest for parity triage (same word order as the HIP library's
 * WebRtcAecmBatch_GetDigest; see include/aecm_batch.h). */
void aecm_oracle_digest(const AecmOracle *o, uint32_t digest[ORC_DIGEST_WORDS]);

void aecm_oracle_get_stats(const AecmOracle *o, uint64_t stats[ORC_STATS]);

/* Exposed for unit tests of the primitives. */
int32_t aecm_oracle_sqrt_floor(int32_t value);                          /* spl.cc:84-105 */
void aecm_oracle_fft128(int16_t re[128], int16_t im[128], int inverse,  /* complex_fft.c */
                        int *scale_out);

#ifdef __cplusplus
}
#endif
#endif /* AECM_ORACLE_H_ */
