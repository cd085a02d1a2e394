/*
 * Drop-in C ABI of the MI355X-native AECM engine: the session interface of the reference
 * (cpuimage/WebRTC_AECM, aecm/echo_control_mobile.h).  Every entry point below replaces the
 * reference function of the same name; the line numbers cite the reference header.
 *
 * Same names, argument meaning, return codes and buffer ownership as the reference, so a caller
 * such as the reference's main.cc:97-147 links against libaecm_mi355x.so unchanged.  The
 * per-block DSP (WebRtcAecm_ProcessBlock, reference aecm/aecm_core_c.cc:368) runs on the GPU as a
 * one-stream launch of the batched HIP kernel; there is no CPU implementation of that path in this
 * library -- if no HIP device is usable, WebRtcAecm_Create returns NULL.
 *
 * For many concurrent streams use the batch extension in aecm_batch.h.
 */
#ifndef AECM_MI355X_ECHO_CONTROL_MOBILE_H_
#define AECM_MI355X_ECHO_CONTROL_MOBILE_H_

#include <stddef.h>
#include <stdint.h>

enum { AecmFalse = 0, AecmTrue };                 /* reference echo_control_mobile.h:18-20 */

/* Errors / warnings (reference echo_control_mobile.h:23-30) */
#define AECM_UNSPECIFIED_ERROR 12000
#define AECM_UNSUPPORTED_FUNCTION_ERROR 12001
#define AECM_UNINITIALIZED_ERROR 12002
#define AECM_NULL_POINTER_ERROR 12003
#define AECM_BAD_PARAMETER_ERROR 12004
#define AECM_BAD_PARAMETER_WARNING 12100

typedef struct {                                  /* reference echo_control_mobile.h:32-35 */
    int16_t cngMode;   /* AecmFalse, AecmTrue (default) */
    int16_t echoMode;  /* 0, 1, 2, 3 (default), 4 */
} AecmConfig;

#ifdef __cplusplus
extern "C" {
#endif

/* reference :46   Allocates an instance (device state for one stream on HIP device 0, or the one named by
 *                 WebRtcAecm_SetDefaultDevice of aecm_batch.h).  NULL on failure (including "no usable GPU"). */
void *WebRtcAecm_Create(void);

/* reference :55   Releases the instance; NULL is a no-op. */
void WebRtcAecm_Free(void *aecmInst);

/* reference :70   sampFreq 8000 or 16000.  0 ok, -1 NULL instance, 12004 bad rate, 12000 device error. */
int32_t WebRtcAecm_Init(void *aecmInst, int32_t sampFreq);

/* reference :87   Buffers 80 or 160 far-end samples.  0 / -1 / 12003 / 12002 / 12004. */
int32_t WebRtcAecm_BufferFarend(void *aecmInst, const int16_t *farend, size_t nrOfSamples);

/* reference :106  The error WebRtcAecm_BufferFarend would return, without buffering. */
int32_t WebRtcAecm_GetBufferFarendError(void *aecmInst, const int16_t *farend, size_t nrOfSamples);

/* reference :135  Processes 80 or 160 near-end samples.  nearendClean may be NULL.  out may alias
 *                 the input.  msInSndCardBuf is clamped to [0, 500] with warning 12100. */
int32_t WebRtcAecm_Process(void *aecmInst, const int16_t *nearendNoisy, const int16_t *nearendClean,
                           int16_t *out, size_t nrOfSamples, int16_t msInSndCardBuf);

/* reference :156  cngMode in {0,1}, echoMode in 0..4 (by value, as in the reference). */
int32_t WebRtcAecm_set_config(void *aecmInst, AecmConfig config);

/* reference :172 / :191 / :202  Stored echo channel import / export; size must be 130 bytes. */
int32_t WebRtcAecm_InitEchoPath(void *aecmInst, const void *echo_path, size_t size_bytes);
int32_t WebRtcAecm_GetEchoPath(void *aecmInst, void *echo_path, size_t size_bytes);
size_t WebRtcAecm_echo_path_size_bytes(void);

#ifdef __cplusplus
}
#endif
#endif /* AECM_MI355X_ECHO_CONTROL_MOBILE_H_ */
