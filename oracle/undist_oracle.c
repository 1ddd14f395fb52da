/*
 * undist_oracle.c — CPU restatement of GSLAM's image undistortion loops.  TEST INFRASTRUCTURE ONLY.
 * PINNED: tests/test_undist_oracle.py compares this file with the reference's own UndistorterImpl::undistort /
 * undistortFast compiled from /root/reference (oracle/_ref, ref_undist_*), on tables built by the reference's own
 * prepareReMap, and with committed golden vectors (tests/golden/undist_reference.npz).
 *
 * Follows GSLAM/core/Undistorter.h:
 *   :271-348  undistort: out = (uchar)(p[i0]*c0 + p[i1]*c1 + p[i2]*c2 + p[i3]*c3), float products summed left to
 *             right, truncating store; 1 channel: 0 where remapX < 0; n channels: only where remapX > 0
 *   :206-268  undistortFast: nearest source pixel remapFast; 1 channel: only where remapFast > 0; 3 channels: every
 *             pixel (the reference reads index -1 for unmapped pixels: undefined, not restated); other: remapX > 0
 * The remap tables (remapX, remapFast, remapIdx[4], remapCoef[4] per output pixel, :120-203) are inputs: they are
 * host-side double-precision camera maths that stay on the host (GSLAM::Camera), as in the reference.
 * Pixels the reference leaves UNWRITTEN (uninitialised memory there) are written as 0 here and on the GPU, and
 * reported in `written` so parity is checked only where the reference defines a value.
 */
#include <stdint.h>
#include <string.h>

/* n_in = pixels of the input image.  The reference's tables address (yyi+1)*w_in + xxi + 1 even when the source
 * position lies in the last row / column (Undistorter.h:184-187), i.e. it reads past the image there (undefined).
 * Indices >= n_in are clamped to n_in - 1 here and on the GPU; `written` is cleared for such pixels in bilinear
 * mode so that parity is only asserted where the reference is defined. */
void oracle_undistort(const uint8_t* img, int channels, int n_in, int n_out, const float* remapX, const int32_t* remapFast,
                      const int32_t* remapIdx, const float* remapCoef, int fast, uint8_t* out, uint8_t* written) {
  memset(out, 0, (size_t)n_out * channels);
  memset(written, 0, (size_t)n_out);
  for (int i = 0; i < n_out; ++i) {
    if (fast) {
      int ok = channels == 1 ? remapFast[i] > 0 : (channels == 3 ? remapFast[i] >= 0 : remapX[i] > 0);
      if (!ok) continue;
      memcpy(out + (size_t)i * channels, img + (size_t)remapFast[i] * channels, channels);
      written[i] = 1;
    } else {
      int32_t id[4];
      int clamped = 0;
      for (int t = 0; t < 4; ++t) {
        id[t] = remapIdx[4 * (size_t)i + t];
        if (id[t] >= n_in) { id[t] = n_in - 1; clamped = 1; }
      }
      const float* co = remapCoef + 4 * (size_t)i;
      if (channels == 1) {
        written[i] = 1;
        if (remapX[i] < 0) continue; /* stays 0 */
        float v = img[id[0]] * co[0] + img[id[1]] * co[1] + img[id[2]] * co[2] + img[id[3]] * co[3];
        out[i] = (uint8_t)v;
        if (clamped) written[i] = 0;
      } else {
        if (!(remapX[i] > 0)) continue;
        for (int j = 0; j < channels; ++j) {
          float v = img[id[0] * channels + j] * co[0] + img[id[1] * channels + j] * co[1] +
                    img[id[2] * channels + j] * co[2] + img[id[3] * channels + j] * co[3];
          out[(size_t)i * channels + j] = (uint8_t)v;
        }
        written[i] = clamped ? 0 : 1;
      }
    }
  }
}
