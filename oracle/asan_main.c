#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
int orc_frame_features(const void*, const void*, int, int, int, int, double, double, const float*, float*, double*);
uint64_t orc_sse_plane(const void*, int, const void*, int, int, int, int);
double orc_ssim_plane(const void*, int, const void*, int, int, int, int);
int main(void) {
  const int sizes[][2] = {{16,16},{33,17},{64,48},{97,33}};
  for (int s = 0; s < 4; ++s) {
    int w = sizes[s][0], h = sizes[s][1];
    uint8_t *r = malloc(w*h), *d = malloc(w*h); float *b = malloc(sizeof(float)*w*h), *b2 = malloc(sizeof(float)*w*h);
    for (int i = 0; i < w*h; ++i) { r[i] = (uint8_t)(i*37 % 251); d[i] = (uint8_t)((i*37 % 251) ^ (i % 7)); }
    double f[17];
    if (orc_frame_features(r, d, w, 8, w, h, 100.0, 100.0, NULL, b, f)) return 1;
    if (orc_frame_features(d, r, w, 8, w, h, 1.0, 1.0, b, b2, f)) return 1;
    printf("%dx%d vif0 %.6f adm0 %.6f motion %.4f sse %llu ssim %.6f\n", w, h, f[0]/f[4], f[8]/f[12], f[16],
           (unsigned long long)orc_sse_plane(r, w, d, w, 8, w, h), orc_ssim_plane(r, w, d, w, 8, w, h));
    free(r); free(d); free(b); free(b2);
  }
  return 0;
}
