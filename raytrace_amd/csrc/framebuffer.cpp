// framebuffer.cpp -- headless image output.
//
// WriteBmp32 produces, byte for byte, what cpu_raycast/bmp.c:6-56 (printImage) writes for
// the same pixels: "BM", file size, 4 zero bytes, data offset 58; a 40-byte info header
// (width, height, 1 plane, 32 bpp, no compression, both resolutions 0xFFFFFF, zero palette
// counts); 4 stray zero bytes; then rows bottom-up, 4 bytes per pixel in memory order.
// The reference opens its file in text mode (cpu_raycast/raycast.cpp:58); this writer is
// binary so 0x0A bytes survive on every platform.
#include "host_scene.h"

#include <cstdio>

namespace rt {

namespace {
void Put32(FILE *f, int32_t v) { std::fwrite(&v, 4, 1, f); }
void Put16(FILE *f, int16_t v) { std::fwrite(&v, 2, 1, f); }
}  // namespace

bool WriteBmp32(const char *path, const uint8_t *bgra, int width, int height)
{
  FILE *f = std::fopen(path, "wb");
  if (!f) return false;
  const int32_t row = width * 4;  // always a multiple of 4: no row padding
  const int32_t data_start = 14 + 40 + 4;
  std::fwrite("BM", 2, 1, f);
  Put32(f, row * height + data_start);
  Put32(f, 0);
  Put32(f, data_start);
  Put32(f, 40);
  Put32(f, width);
  Put32(f, height);
  Put16(f, 1);
  Put16(f, 32);
  Put32(f, 0);
  Put32(f, 0);
  Put32(f, 0xFFFFFF);
  Put32(f, 0xFFFFFF);
  Put32(f, 0);
  Put32(f, 0);
  Put32(f, 0);
  for (int y = height - 1; y >= 0; --y) std::fwrite(bgra + static_cast<size_t>(y) * row, row, 1, f);
  const bool ok = !std::ferror(f);
  return std::fclose(f) == 0 && ok;
}

// Little-endian RGB PFM (scale -1.0); alpha is dropped, rows bottom-up as the format requires.
bool WritePfm(const char *path, const float *rgba, int width, int height)
{
  FILE *f = std::fopen(path, "wb");
  if (!f) return false;
  std::fprintf(f, "PF\n%d %d\n-1.0\n", width, height);
  for (int y = height - 1; y >= 0; --y)
    for (int x = 0; x < width; ++x) std::fwrite(rgba + (static_cast<size_t>(y) * width + x) * 4, sizeof(float), 3, f);
  const bool ok = !std::ferror(f);
  return std::fclose(f) == 0 && ok;
}

}  // namespace rt
