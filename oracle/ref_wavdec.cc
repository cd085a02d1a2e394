// TEST INFRASTRUCTURE ONLY.  A door onto the WAV reader the reference CLI uses (dr_wav.h, vendored in the reference tree and
// reached through -I$(REF); no reference code in this file): decodes a file exactly as main.cc's wavRead_int16 does
// (drwav_open_file_and_read_pcm_frames_s16, main.cc:39-54) and writes the int16 samples raw to stdout's file argument.
//   ref_wavdec in.wav out.raw      prints "<channels> <rate> <frames>"
#define DR_WAV_IMPLEMENTATION
#include "dr_wav.h"

#include <stdint.h>
#include <stdio.h>

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    unsigned int channels = 0, rate = 0;
    drwav_uint64 frames = 0;
    int16_t *s = drwav_open_file_and_read_pcm_frames_s16(argv[1], &channels, &rate, &frames, NULL);
    if (!s) { printf("0 0 0\n"); return 1; }
    FILE *f = fopen(argv[2], "wb");
    if (!f) return 3;
    fwrite(s, sizeof(int16_t), (size_t)(frames * channels), f);
    fclose(f);
    printf("%u %u %llu\n", channels, rate, (unsigned long long)frames);
    drwav_free(s, NULL);
    return 0;
}
