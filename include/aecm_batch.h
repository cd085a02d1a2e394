/*
 * Batch extension of the AECM C ABI: S independent AECM block streams resident on one MI355X,
 * one wavefront per stream.  Not present in the reference; it is the data-parallel form of the
 * reference's block-level core interface (aecm/aecm_core.h:149-239):
 *
 *   WebRtcAecmBatch_Create/Free      <->  WebRtcAecm_CreateCore / FreeCore   (aecm_core.h:149,198)
 *   WebRtcAecmBatch_Init             <->  WebRtcAecm_InitCore                (aecm_core.h:166)
 *                                         + default set_config (echo_control_mobile.cc:183-188)
 *   WebRtcAecmBatch_set_config       <->  core part of WebRtcAecm_set_config (echo_control_mobile.cc:410-479)
 *   WebRtcAecmBatch_Control          <->  WebRtcAecm_Control                 (aecm_core.h:200)
 *   WebRtcAecmBatch_ProcessBlocks    <->  T x WebRtcAecm_ProcessBlock per stream (aecm_core.h:235)
 *   WebRtcAecmBatch_InitEchoPath / GetEchoPath <-> WebRtcAecm_InitEchoPathCore (aecm_core.h:187) /
 *                                         WebRtcAecm_GetEchoPath (echo_control_mobile.h:191)
 *
 * Return codes follow echo_control_mobile.h: 0 ok, -1 NULL handle, 12002 not initialised,
 * 12003 NULL data pointer, 12004 bad parameter, 12000 device (HIP) failure.
 *
 * Audio layout: sample i of block b of stream s is at base[s*stream_stride + b*block_stride + i]
 * (strides in int16 elements).  Stream-major "files" are (stream_stride, block_stride) = (T*64, 64);
 * a tick-major server layout [T][S][64] is (64, S*64).  Each wavefront reads/writes whole 128-byte
 * lines either way.
 */
#ifndef AECM_MI355X_BATCH_H_
#define AECM_MI355X_BATCH_H_

#include <stddef.h>
#include <stdint.h>

#include "echo_control_mobile.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct AecmBatch AecmBatch;

enum { AECM_BATCH_DIGEST_WORDS = 24 };
enum { AECM_KERNEL_SAFE = 0, AECM_KERNEL_FAST = 1 };

/* Allocates device state for num_streams streams on HIP device device_id.  NULL on failure. */
AecmBatch *WebRtcAecmBatch_Create(int32_t num_streams, int32_t device_id);
void WebRtcAecmBatch_Free(AecmBatch *b);

int32_t WebRtcAecmBatch_num_streams(const AecmBatch *b);

/* (Re)initialises every stream for sampFreq in {8000, 16000}, default config cng=1, echoMode=3. */
int32_t WebRtcAecmBatch_Init(AecmBatch *b, int32_t sampFreq);
/* Applies config to streams [first, first+count); count < 0 means "to the end". */
int32_t WebRtcAecmBatch_set_config(AecmBatch *b, AecmConfig config, int32_t first, int32_t count);
/* WebRtcAecm_Control (aecm_core.cc:477-482) for streams [first, first+count).  Deviation from the reference, on purpose:
 * the reference stores both arguments unchecked (narrowed to int16_t) and later uses fixedDelay as an offset into the
 * 100-slot far history (aecm_core.cc:157-172).  Here values the core could not hold or index with are refused with
 * AECM_BAD_PARAMETER_ERROR instead of being narrowed silently: fixed_delay must lie in [-32768, 100) (negative = "use
 * the delay estimator", as in the reference), nlp_flag in [-32768, 32767]. */
int32_t WebRtcAecmBatch_Control(AecmBatch *b, int32_t fixed_delay, int32_t nlp_flag, int32_t first, int32_t count);

/* Runs num_blocks consecutive 64-sample blocks of every stream.  All four pointers are DEVICE
 * pointers (near_clean may be NULL).  Asynchronous on the engine's HIP stream. */
int32_t WebRtcAecmBatch_ProcessBlocks(AecmBatch *b, const int16_t *far_dev, const int16_t *near_dev,
                                      const int16_t *near_clean_dev, int16_t *out_dev, int64_t stream_stride,
                                      int64_t block_stride, int32_t num_blocks);
/* Same with HOST pointers: copies in, runs, copies out, synchronises. */
int32_t WebRtcAecmBatch_ProcessBlocksHost(AecmBatch *b, const int16_t *far_host, const int16_t *near_host,
                                          const int16_t *near_clean_host, int16_t *out_host, int64_t stream_stride,
                                          int64_t block_stride, int32_t num_blocks);
/* Whole recordings as sessions (batched form of the reference CLI's loop, main.cc:105-143): every
 * stream is treated as a freshly initialised WebRtcAecm_* session that receives num_calls pairs of
 * WebRtcAecm_BufferFarend(far, samples_per_call) + WebRtcAecm_Process(near, near_clean, out,
 * samples_per_call, msInSndCardBuf) (near_clean may be NULL, reference echo_control_mobile.h:135-140).  The session wrapper and frame adapter of the reference
 * (echo_control_mobile.cc:236-408, aecm_core.cc:501-572) only move samples, so they are run once on
 * the host in the index domain and applied to all streams as a device-side gather / scatter around
 * WebRtcAecmBatch_ProcessBlocks.  Call after WebRtcAecmBatch_Init (+ set_config) on fresh streams.
 * far/near/near_clean/out: [S][stream_stride] with stream_stride >= num_calls * samples_per_call; samples beyond
 * that are not touched.  Returns 0 or AECM_BAD_PARAMETER_WARNING exactly as each session would.
 * Synchronous. */
int32_t WebRtcAecmBatch_ProcessRecordings(AecmBatch *b, const int16_t *far_dev, const int16_t *near_dev,
                                          const int16_t *near_clean_dev, int16_t *out_dev, int64_t stream_stride,
                                          int32_t samples_per_call, int32_t num_calls, int16_t msInSndCardBuf);
int32_t WebRtcAecmBatch_ProcessRecordingsHost(AecmBatch *b, const int16_t *far_host, const int16_t *near_host,
                                              const int16_t *near_clean_host, int16_t *out_host, int64_t stream_stride,
                                              int32_t samples_per_call, int32_t num_calls, int16_t msInSndCardBuf);
int32_t WebRtcAecmBatch_Synchronize(AecmBatch *b);

/* Duration of the most recent timed ProcessBlocks kernel, from HIP events recorded around the launch on
 * the engine's own stream (waits for it to finish).
 * Timed: every launch made by WebRtcAecmBatch_ProcessBlocks and by the staged form of ProcessBlocksHost /
 * ProcessRecordings*.  NOT timed (latency paths that would pay for the event pair): host calls of at most 64 KiB of
 * audio, which run the kernel directly on a pinned mapped buffer (the whole single-session WebRtcAecm_* ABI; switch
 * off with AECM_HOST_MAPPED=0), and the session ticks (WebRtcAecmSessions_Tick*).  For those, GetTimers / GetLastLaunchMs
 * keep reporting the last timed launch. */
int32_t WebRtcAecmBatch_GetLastLaunchMs(AecmBatch *b, float *ms);
/* Sum of the timed kernel durations (see above) since the last call to WebRtcAecmBatch_ResetTimers and their count. */
int32_t WebRtcAecmBatch_GetTimers(AecmBatch *b, double *total_ms, int64_t *launches);
int32_t WebRtcAecmBatch_ResetTimers(AecmBatch *b);

/* Per-stream stored echo channel (65 x int16 = 130 bytes), as echo_control_mobile.h:172,191. */
int32_t WebRtcAecmBatch_InitEchoPath(AecmBatch *b, int32_t stream, const void *echo_path, size_t size_bytes);
int32_t WebRtcAecmBatch_GetEchoPath(AecmBatch *b, int32_t stream, void *echo_path, size_t size_bytes);

/* Full snapshot of one stream's device state (checkpoint / migration between batches or GPUs): a 32-byte
 * header (magic, layout version, sampling rate, layout sizes) + lane-vector words, scalars and far-spectrum
 * history, WebRtcAecmBatch_state_size_bytes() bytes.  A stream restored with ImportState continues bit-exactly
 * where the exported one stopped.  ImportState refuses (AECM_BAD_PARAMETER_ERROR) a blob whose header is not
 * this build's layout or whose index-like fields are out of range.  Importing a stream of the other sampling
 * rate is allowed (the rate is part of the state) but disables WebRtcAecmBatch_ProcessRecordings
 * (AECM_UNSUPPORTED_FUNCTION_ERROR) until the next WebRtcAecmBatch_Init. */
size_t WebRtcAecmBatch_state_size_bytes(void);
int32_t WebRtcAecmBatch_ExportState(AecmBatch *b, int32_t stream, void *state, size_t size_bytes);
int32_t WebRtcAecmBatch_ImportState(AecmBatch *b, int32_t stream, const void *state, size_t size_bytes);

/* The same for streams [first, first + count) at once: `states` = count snapshots of WebRtcAecmBatch_state_size_bytes()
 * each, back to back (size_bytes = count x that).  One gather (scatter) launch on the device and one copy per 8 192 streams
 * instead of three blocking copies per stream.  The *Device forms take anything the device can address: device memory
 * (e.g. the staging buffer of a peer-to-peer migration), or the alias of a caller-owned host buffer registered with
 * WebRtcAecmBatch_RegisterHostBuffer -- the kernels then write (read) the snapshots in place over the link.
 * ImportStates is all or nothing: every snapshot is validated exactly like ImportState validates one (on the device for
 * the *Device form) before any stream is touched; AECM_BAD_PARAMETER_ERROR if one of them may not be run on. */
int32_t WebRtcAecmBatch_ExportStates(AecmBatch *b, int32_t first, int32_t count, void *states_host, size_t size_bytes);
int32_t WebRtcAecmBatch_ImportStates(AecmBatch *b, int32_t first, int32_t count, const void *states_host, size_t size_bytes);
/* (The *Device forms: states_dev must be 4-byte aligned -- AECM_BAD_PARAMETER_ERROR otherwise -- and, for ImportStatesDevice, must not
 * change while the call runs: the blobs are validated by one launch and written into the streams by a second one.) */
int32_t WebRtcAecmBatch_ExportStatesDevice(AecmBatch *b, int32_t first, int32_t count, void *states_dev, size_t size_bytes);
int32_t WebRtcAecmBatch_ImportStatesDevice(AecmBatch *b, int32_t first, int32_t count, const void *states_dev, size_t size_bytes);

/* 24-word digest of one stream's complete state (canonical order: oracle/aecm_oracle.c). */
int32_t WebRtcAecmBatch_GetDigest(AecmBatch *b, int32_t stream, uint32_t digest[AECM_BATCH_DIGEST_WORDS]);

/* ---- Streaming batch of sessions --------------------------------------------------------------------
 * S independent WebRtcAecm_* sessions that share one call pattern (a media server's 10 ms clock):
 * every WebRtcAecmSessions_Tick is, for each stream s,
 *     WebRtcAecm_BufferFarend(inst_s, far[s], nrOfSamples);
 *     WebRtcAecm_Process(inst_s, near[s], near_clean ? near_clean[s] : NULL, out[s], nrOfSamples, msInSndCardBuf);
 * (reference echo_control_mobile.h:87,135) with identical results and return code.  Audio is held in
 * per-stream rings in HBM; the reference's jitter buffer / start-up gating / delay compensation /
 * 80->64 re-blocking run on the device, per session, as position arithmetic on per-session state
 * (csrc/aecm_flow_plan.h), so sessions need not have anything in common but the tick.
 * far/near/near_clean/out: [S][stream_stride] int16, device pointers (Tick) or host pointers (TickHost);
 * near_clean may be NULL, but a batch must either always or never pass it (the clean ring is only kept
 * up to date by ticks that carry it). */
typedef struct AecmSessions AecmSessions;
AecmSessions *WebRtcAecmSessions_Create(int32_t num_streams, int32_t device_id);
void WebRtcAecmSessions_Free(AecmSessions *s);
int32_t WebRtcAecmSessions_Init(AecmSessions *s, int32_t sampFreq);
int32_t WebRtcAecmSessions_set_config(AecmSessions *s, AecmConfig config);
int32_t WebRtcAecmSessions_Tick(AecmSessions *s, const int16_t *far_dev, const int16_t *near_dev,
                                const int16_t *near_clean_dev, int16_t *out_dev, int64_t stream_stride, size_t nrOfSamples,
                                int16_t msInSndCardBuf);
int32_t WebRtcAecmSessions_TickHost(AecmSessions *s, const int16_t *far_host, const int16_t *near_host,
                                    const int16_t *near_clean_host, int16_t *out_host, int64_t stream_stride,
                                    size_t nrOfSamples, int16_t msInSndCardBuf);

/* The same with one msInSndCardBuf per session (host array, S entries): session s runs
 *     WebRtcAecm_Process(inst_s, ..., msInSndCardBuf_host[s]).
 * codes_host (S entries, may be NULL) receives each session's return code; the function returns 0 or the
 * first non-zero code.  Any number of distinct msInSndCardBuf histories per object; Tick and TickPerSession may
 * be mixed. */
int32_t WebRtcAecmSessions_TickPerSession(AecmSessions *s, const int16_t *far_dev, const int16_t *near_dev,
                                          const int16_t *near_clean_dev, int16_t *out_dev, int64_t stream_stride,
                                          size_t nrOfSamples, const int16_t *msInSndCardBuf_host, int32_t *codes_host);
int32_t WebRtcAecmSessions_TickPerSessionHost(AecmSessions *s, const int16_t *far_host, const int16_t *near_host,
                                              const int16_t *near_clean_host, int16_t *out_host, int64_t stream_stride,
                                              size_t nrOfSamples, const int16_t *msInSndCardBuf_host, int32_t *codes_host);
/* The same with per-session call flags (host array, S entries): AECM_SESSION_NO_FAREND = this session gets NO
 * WebRtcAecm_BufferFarend call in this tick (far-end underrun; its WebRtcAecm_Process then replays the previous
 * far frame, reference echo_control_mobile.cc:369-380) -- its far row is ignored. */
enum {
    AECM_SESSION_NO_FAREND = 1,
    /* 160-sample ticks only: this session makes TWO WebRtcAecm_BufferFarend + WebRtcAecm_Process call pairs of 80 samples
     * in this tick (first half, then second half of its rows) instead of one pair of 160 samples -- sessions with
     * different call sizes in one object (the reference treats the two cadences differently,
     * echo_control_mobile.cc:282-283, 384-385).  codes_host then receives the first non-zero code of the two calls. */
    AECM_SESSION_SPLIT_CALLS = 2
};
int32_t WebRtcAecmSessions_TickFlags(AecmSessions *s, const int16_t *far_dev, const int16_t *near_dev,
                                     const int16_t *near_clean_dev, int16_t *out_dev, int64_t stream_stride, size_t nrOfSamples,
                                     const int16_t *msInSndCardBuf_host, const uint8_t *flags_host, int32_t *codes_host);
int32_t WebRtcAecmSessions_TickFlagsHost(AecmSessions *s, const int16_t *far_host, const int16_t *near_host,
                                         const int16_t *near_clean_host, int16_t *out_host, int64_t stream_stride,
                                         size_t nrOfSamples, const int16_t *msInSndCardBuf_host, const uint8_t *flags_host,
                                         int32_t *codes_host);

/* The asynchronous form: the tick's two launches are enqueued on the object's own HIP stream and the call returns
 * without waiting for them, so a caller that keeps its audio on the device can issue tick t + 1 (and its own producer /
 * consumer kernels) while tick t runs.  Device pointers only (or device aliases of registered host buffers, below).
 *   msInSndCardBuf_host == NULL : every session gets msInSndCardBuf; else S entries, read before the call returns
 *   flags_host                  : NULL or S entries (needs msInSndCardBuf_host), read before the call returns
 *   codes_host                  : NULL or S entries, written before the call returns (the codes do not depend on the device)
 *   wait_hip_event (hipEvent_t) : NULL, or an event the tick's kernels wait for (the caller's producer of far / near)
 *   done_hip_event (hipEvent_t) : NULL, or an event recorded behind the tick (the caller's consumer of out waits for it)
 * The far / near / out buffers must stay untouched until the tick has run (done_hip_event or WebRtcAecmSessions_Synchronize).
 * Every other WebRtcAecmSessions_* call is ordered behind the ticks enqueued so far; the synchronous Tick forms
 * wait for everything. */
int32_t WebRtcAecmSessions_TickAsync(AecmSessions *s, const int16_t *far_dev, const int16_t *near_dev, const int16_t *near_clean_dev,
                                     int16_t *out_dev, int64_t stream_stride, size_t nrOfSamples, int16_t msInSndCardBuf,
                                     const int16_t *msInSndCardBuf_host, const uint8_t *flags_host, int32_t *codes_host,
                                     void *wait_hip_event, void *done_hip_event);
/* Far-end bursts.  The reference lets a caller make ANY number of WebRtcAecm_BufferFarend calls between two
 * WebRtcAecm_Process calls (echo_control_mobile.h:87,135) -- what a jittery network produces: no far frame for a tick or
 * two, then several at once.  Each call runs the delay compensation once the session is past its start-up phase and then
 * appends to the 4 000-sample jitter buffer, which drops what does not fit (echo_control_mobile.cc:215-234, 575-594;
 * ring_buffer.c:142-170).  WebRtcAecmSessions_BufferFarend makes, for every session s,
 *     for (c = 0; c < (calls_host ? calls_host[s] : calls); ++c) WebRtcAecm_BufferFarend(inst_s, far[s] + c * nrOfSamples, nrOfSamples);
 * far: [S][stream_stride] int16 with stream_stride >= calls * nrOfSamples; calls in [0, 255]; calls_host: NULL or S entries
 * <= calls, read before the call returns.  Returns what the reference's call returns (0; AECM_NULL_POINTER_ERROR,
 * AECM_UNINITIALIZED_ERROR, AECM_BAD_PARAMETER_ERROR for the arguments, in its order).
 * WebRtcAecmSessions_Process is the other half: every session's WebRtcAecm_Process WITHOUT a WebRtcAecm_BufferFarend (a tick
 * in which every session carries AECM_SESSION_NO_FAREND); msInSndCardBuf_host / codes_host as in TickPerSession, both may be
 * NULL.  With the two, k = 0, 1, 2, ... far calls per near call and session are expressed as: BufferFarend with
 * calls_host[s] = k_s, then Process.  The Tick* forms remain the one-launch shape of the common k = 1 (TickFlags: k in
 * {0, 1} per session); in TickFlags / TickAsync far may be NULL when every session carries AECM_SESSION_NO_FAREND.
 * BufferFarend (device pointers) and BufferFarendAsync are enqueued on the object's stream like ticks; BufferFarend and
 * BufferFarendHost wait for it, BufferFarendAsync does not (events as in TickAsync; the far rows must stay untouched
 * until the launch has run). */
int32_t WebRtcAecmSessions_BufferFarend(AecmSessions *s, const int16_t *far_dev, int64_t stream_stride, size_t nrOfSamples, int32_t calls,
                                        const uint8_t *calls_host);
int32_t WebRtcAecmSessions_BufferFarendHost(AecmSessions *s, const int16_t *far_host, int64_t stream_stride, size_t nrOfSamples, int32_t calls,
                                            const uint8_t *calls_host);
int32_t WebRtcAecmSessions_BufferFarendAsync(AecmSessions *s, const int16_t *far_dev, int64_t stream_stride, size_t nrOfSamples, int32_t calls,
                                             const uint8_t *calls_host, void *wait_hip_event, void *done_hip_event);
int32_t WebRtcAecmSessions_Process(AecmSessions *s, const int16_t *near_dev, const int16_t *near_clean_dev, int16_t *out_dev, int64_t stream_stride,
                                   size_t nrOfSamples, int16_t msInSndCardBuf, const int16_t *msInSndCardBuf_host, int32_t *codes_host);
int32_t WebRtcAecmSessions_ProcessHost(AecmSessions *s, const int16_t *near_host, const int16_t *near_clean_host, int16_t *out_host,
                                       int64_t stream_stride, size_t nrOfSamples, int16_t msInSndCardBuf, const int16_t *msInSndCardBuf_host,
                                       int32_t *codes_host);
int32_t WebRtcAecmSessions_Synchronize(AecmSessions *s);
/* AECM_KERNEL_FAST (default) / AECM_KERNEL_SAFE for the object's block engine.  The tick kernel is built on the fast
 * primitives only: with the safe variant selected, ticks return AECM_UNSUPPORTED_FUNCTION_ERROR (and change nothing). */
int32_t WebRtcAecmSessions_SetKernelVariant(AecmSessions *s, int32_t variant);

/* Zero-copy host audio.  A caller-owned host buffer is pinned and mapped into the device's address space once
 * (hipHostRegister); *device_alias is then a device pointer that every *_dev argument of this header accepts
 * (WebRtcAecmBatch_ProcessBlocks, WebRtcAecmSessions_Tick / TickAsync ...): the kernels read the samples and write the
 * output in place over the link -- no staging copies, no pageable-memory transfers (the *Host forms stage through
 * pageable copies: 1.44 ms per tick of 65 536 sessions against the numbers in INTEGRATION.md for this form). */
int32_t WebRtcAecmBatch_RegisterHostBuffer(int32_t device_id, void *host, size_t size_bytes, void **device_alias);
int32_t WebRtcAecmBatch_UnregisterHostBuffer(int32_t device_id, void *host);

/* Per-session control while the other sessions keep running (a media server recycling one slot when a call
 * ends).  `session` in [0, S); same return codes as the single-session functions they mirror:
 *   InitSession           <-> WebRtcAecm_Init(inst_s, same sampFreq)   (echo_control_mobile.h:70): fresh core state,
 *                             fresh jitter buffer / start-up phase, default config (cng on, echoMode 3)
 *   set_config_session    <-> WebRtcAecm_set_config(inst_s, config)   (:156)
 *   InitEchoPath / GetEchoPath <-> WebRtcAecm_InitEchoPath / GetEchoPath (:172, :191), 130 bytes
 * InitSession and set_config_session are kernel launches ordered before the next tick on the object's stream (no
 * synchronisation with the host); InitEchoPath / GetEchoPath synchronise. */
int32_t WebRtcAecmSessions_InitSession(AecmSessions *s, int32_t session);
int32_t WebRtcAecmSessions_set_config_session(AecmSessions *s, int32_t session, AecmConfig config);
int32_t WebRtcAecmSessions_InitEchoPath(AecmSessions *s, int32_t session, const void *echo_path, size_t size_bytes);
int32_t WebRtcAecmSessions_GetEchoPath(AecmSessions *s, int32_t session, void *echo_path, size_t size_bytes);

/* Snapshot of ONE live session (checkpoint; migration of a call into another AecmSessions object, on another GPU):
 * everything the reference keeps per instance -- AecMobile's wrapper members and its far-end jitter buffer
 * (echo_control_mobile.cc:42-79), the core's frame buffers and the core state (aecm_core.h:41-141) -- in a form that does
 * not depend on the object it was taken from.  A session imported into any slot of any object of the same sampling rate
 * continues bit-exactly where the exported one stopped, whatever the two objects' ages; the exporting object is not
 * changed.  ImportSession refuses (AECM_BAD_PARAMETER_ERROR, nothing changed) a snapshot of another layout or rate, one
 * whose wrapper state could not have been reached (buffer fills, pending samples, flags), and a core state ImportState
 * would refuse.  Both calls wait for the ticks enqueued so far. */
size_t WebRtcAecmSessions_session_size_bytes(void);
int32_t WebRtcAecmSessions_ExportSession(AecmSessions *s, int32_t session, void *snapshot, size_t size_bytes);
int32_t WebRtcAecmSessions_ImportSession(AecmSessions *s, int32_t session, const void *snapshot, size_t size_bytes);

/* AECM_KERNEL_FAST (default) or AECM_KERNEL_SAFE cross-lane primitives. */
int32_t WebRtcAecmBatch_SetKernelVariant(AecmBatch *b, int32_t variant);

/* How a ProcessBlocks launch is scheduled on the device; results do not depend on it.  A launch of more streams than
 * the pipelined form takes (4 096 on an MI355X) is cut into chunks of chunk_blocks blocks that resident wavefronts
 * claim in order from a queue, so that all streams advance together and the launch does not end at low occupancy
 * (default: 128 blocks; the default is quartered, to at least 8, while every stream's wavefront is resident at once -- a
 * chunk_blocks set through this call is taken as it is).  (Shorthand for the queue_* fields of AecmLaunchPolicy, below.)
 * chunk_blocks = 0: one wavefront keeps one stream for the whole launch, always.  min_streams < 0 (default): the
 * threshold above; >= 0: the queue form above that many streams (diagnostics / tests). */
int32_t WebRtcAecmBatch_SetLaunchChunking(AecmBatch *b, int32_t chunk_blocks, int32_t min_streams);
/* Launches the chip holds at once (<= 16 streams per compute unit = 4 096 streams on an MI355X; fast variant, no clean
 * near-end input) run pipelined: a workgroup serves up to four streams with the state-independent transforms of a block in
 * "front" wavefronts of their own, one block ahead of the rest.  Shape by streams per compute unit: up to 8 (2 048 streams)
 * sixteen wavefronts per workgroup, two workgroups per unit -- the delay estimator one block ahead, the gain half of the block and
 * the inverse transforms one block behind, each in wavefronts of their own; up to 12 (3 072) eight wavefronts (front and "tail"
 * wavefronts of two streams each), three workgroups per unit; above that six (no tail wavefronts), four per unit, the workgroups kept
 * in step through progress feedback on their front wavefronts' issue priority.  Every compute unit gets the shape's full count of
 * workgroups, of one to four streams each, so that unit loads differ by at most one stream.
 * min_streams: the smallest batch that takes this form (default 2; <= 0: never).  Results do not depend on it.
 * By default launches of one or two blocks keep one wavefront per stream (the pipeline's fill and drain steps cost more than
 * they save there); after this call launches of any length from min_streams streams are pipelined.  (Shorthand for
 * pipelined_min_streams / pipelined_min_blocks of AecmLaunchPolicy, below; its pipe_* fields override the shape.) */
int32_t WebRtcAecmBatch_SetLaunchPipelining(AecmBatch *b, int32_t min_streams);
/* Which form a ProcessBlocks launch of num_blocks blocks over the whole batch takes, with (has_clean_input != 0) or
 * without a clean near-end input (for measurement tools that must name
 * the kernel they time): 0 = one wavefront per stream, kernel variants for launches the chip holds at once; 1 = one
 * wavefront per stream, issue priority by phase; 2 = the chunk queue (*chunk_blocks, if not NULL, receives the chunk);
 * 3 = pipelined (*chunk_blocks then receives the shape: the "tail" wavefronts per workgroup, 0 or 2, + 0x100 when the
 * launch balances its workgroups' progress, + 0x200 with four front wavefronts instead of two, + 0x400 when the back
 * wavefronts form the spectra, + 0x800 with delay wavefronts, + 0x1000 with gain wavefronts). */
#define AECM_LAUNCH_RESIDENT 0
#define AECM_LAUNCH_PER_STREAM 1
#define AECM_LAUNCH_CHUNK_QUEUE 2
#define AECM_LAUNCH_PIPELINED 3
int32_t WebRtcAecmBatch_DescribeLaunch(const AecmBatch *b, int32_t num_blocks, int32_t has_clean_input, int32_t *chunk_blocks);
/* The same answer for a batch of num_streams streams (fast variant, the default launch policy) on a device with
 * compute_units compute units -- no device and no batch needed (capacity planning; -1 for a non-positive argument). */
int32_t WebRtcAecmBatch_DescribeLaunchFor(int32_t num_streams, int32_t compute_units, int32_t num_blocks, int32_t has_clean_input,
                                          int32_t *chunk_blocks);

/* The launch policy: every threshold and wish that decides how a ProcessBlocks launch is scheduled, as ONE value.  The library derives
 * it from the device's compute units (WebRtcAecmBatch_DefaultLaunchPolicy does the same without a device) and from what its kernels are
 * built for; a caller may read it, change fields and set it back (per batch).  Results never depend on it.  The shipped library reads
 * no environment variable: experiments set a policy, or use a build with -DAECM_EXPERIMENTS (tools/ab_build.py) whose default policy
 * takes the AECM_* wishes of the environment.
 *   struct_size            sizeof(AecmLaunchPolicy) of the caller's header (the calls refuse another size)
 *   compute_units          of the device (Set refuses a policy made for another count)
 *   queue_chunk_blocks     chunk queue: blocks per item (128); 0 = never the queue: one wavefront keeps one stream for the whole launch
 *   queue_chunk_explicit   0: the default length, quartered (to at least 8) while every stream's wavefront is resident at once; 1: as given
 *   queue_min_streams      the queue above this many streams; -1: above pipelined_max_streams
 *   pipelined_min_streams  the smallest batch whose launches run pipelined (2); larger than the batch: never
 *   pipelined_min_blocks   the shortest launch that does (3: the pipeline's fill and drain steps cost more than they save below)
 *   pipelined_max_streams  the largest batch that does (16 per compute unit: what the chip holds of the widest shape)
 *   resident_waves         wavefronts of the one-wavefront-per-stream kernels the chip holds (28 per compute unit)
 *   rotation_stream_limit  launches of at most this many streams take the kernel variants built for full residency
 *   pipe_*                 wishes for the pipelined form's shape, -1 = by size (tests, experiments): tail wavefronts 0 / 2, front wavefronts
 *                          2 / 4, raw hand-over 0 / 1, delay wavefronts 0 / 2 / 4, gain wavefronts 0 / 4; pipe_spread 1 = every compute
 *                          unit gets the shape's full count of workgroups, of fewer than four streams each where the launch is short of
 *                          streams (0: workgroups of four); pipe_wgs_per_cu > 0 = workgroups of the shape a compute unit takes;
 *                          pipe_rot >= 0 = the slot rotations, pipe_prio >= 0 = the role wavefronts' issue priorities (csrc/aecm_kernels.h:
 *                          PipeShape::rot / prio) */
typedef struct AecmLaunchPolicy {
    int32_t struct_size;
    int32_t compute_units;
    int32_t queue_chunk_blocks, queue_chunk_explicit, queue_min_streams;
    int32_t pipelined_min_streams, pipelined_min_blocks, pipelined_max_streams;
    int32_t resident_waves, rotation_stream_limit;
    int32_t pipe_tail_waves, pipe_front_waves, pipe_raw, pipe_delay_waves, pipe_gain_waves, pipe_spread, pipe_wgs_per_cu, pipe_rot, pipe_prio;
} AecmLaunchPolicy;
int32_t WebRtcAecmBatch_DefaultLaunchPolicy(int32_t compute_units, AecmLaunchPolicy *policy);
int32_t WebRtcAecmBatch_GetLaunchPolicy(const AecmBatch *b, AecmLaunchPolicy *policy);
int32_t WebRtcAecmBatch_SetLaunchPolicy(AecmBatch *b, const AecmLaunchPolicy *policy);
/* What a launch looks like on the device, for capacity planning (no device needed): the form and its detail as DescribeLaunch gives
 * them, the grid, and how the grid quantises on the chip -- rounds_x1000 = 1000 x workgroups / (compute units x workgroups a compute
 * unit holds at once): 1000 = the chip exactly full once; 9140 = nine full rounds and a last one 14 % full (65 536 sessions' tick);
 * and how evenly a pipelined launch loads the compute units (cu_load_evenness_x1000: batches that are multiples of the unit count
 * lose nothing, one stream more loses up to 1 / (streams per unit + 1) of the frame rate).
 * policy == NULL: the default policy of compute_units.  WebRtcAecmSessions_DescribeTick: the same for one tick of num_sessions sessions. */
typedef struct AecmLaunchDescription {
    int32_t form, chunk_blocks, shape;
    int32_t workgroups, waves_per_workgroup, workgroups_per_cu, rounds_x1000;
    /* pipelined launches keep every stream on one compute unit for the whole launch, so the launch ends when the fullest unit does:
     * 1000 x (streams / compute units) / (streams on the fullest unit) -- 1000 when the batch is a multiple of the unit count,
     * 800 for 1 025 streams on 256 units (one unit carries five streams, the average four).  1000 for the other forms (the chunk queue
     * balances dynamically; one wavefront per stream runs in rounds: rounds_x1000). */
    int32_t cu_load_evenness_x1000;
} AecmLaunchDescription;
int32_t WebRtcAecmBatch_DescribeLaunchDetail(const AecmLaunchPolicy *policy, int32_t compute_units, int32_t num_streams, int32_t num_blocks,
                                             int32_t has_clean_input, AecmLaunchDescription *out);
int32_t WebRtcAecmSessions_DescribeTick(int32_t num_sessions, int32_t compute_units, AecmLaunchDescription *out);
/* The HIP device WebRtcAecm_Create (which has no device argument) puts its sessions on from now on; process-wide, default 0. */
int32_t WebRtcAecm_SetDefaultDevice(int32_t device_id);

/* Device self test of the wave primitives on device_id; failures[0..7] must all be 0 afterwards
 * (see webrtc_aecm_amd/csrc/aecm_kernels.h).  exhaustive != 0 checks floor-sqrt on all of [0, 2^31). */
int32_t WebRtcAecmBatch_SelfTest(int32_t device_id, int32_t exhaustive, uint64_t failures[8]);

/* Precondition audit (diagnostics).  The block kernel replaces some of the reference's arithmetic by cheaper
 * instructions where an operand range is provable (24-bit multiplies, int16 narrowings that are the identity).
 * libaecm_mi355x_checked.so is the same library built with -DAECM_CHECKED: its kernels verify every such claim at
 * run time and count violations -- counters[0] 24-bit multiply operands, counters[1] int16 narrowings -- while still
 * producing exact results.  The shipped library carries no checks and returns AECM_UNSUPPORTED_FUNCTION_ERROR. */
int32_t WebRtcAecmBatch_GetCheckCounters(int32_t device_id, uint64_t counters[2], int32_t reset);

/* Diagnostics: `count` independent 128-point transforms of the block kernel's own FFT code, on host
 * data in natural order (transform k: data[k*256 .. +128) = re, [.. +256) = im, in place).  variant 0 =
 * forward of real input (im ignored), 1 = forward complex, 2 = inverse (reference
 * WebRtcSpl_ComplexBitReverse + ComplexFFT / ComplexIFFT, aecm/complex_fft.c:181-491, mode 1);
 * scales[k] receives the inverse transform's return value.  Outputs the block path never consumes
 * come back as 0: forward -> im of bins >= 64, inverse -> im.  kernel_variant as in SetKernelVariant. */
int32_t WebRtcAecmBatch_DebugFft128(int32_t device_id, int16_t *data_host, int32_t *scales_host, int32_t variant,
                                    int32_t kernel_variant, int32_t count);

/* Name, CU count and clock of the device the library would use (diagnostics). */
int32_t WebRtcAecmBatch_DeviceInfo(int32_t device_id, char *name, size_t name_len, int32_t *compute_units,
                                   int32_t *clock_khz);
/* Its PCI bus id ("0000:05:00.0"; bus_id_len >= 13): which physical GPU a rank of a multi-GPU run really had. */
int32_t WebRtcAecmBatch_DevicePciBusId(int32_t device_id, char *bus_id, size_t bus_id_len);

#ifdef __cplusplus
}
#endif
#endif /* AECM_MI355X_BATCH_H_ */
