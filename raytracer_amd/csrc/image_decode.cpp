// image_decode.cpp - colour-map files to RGBA8, row 0 = top (what the reference gets
// from image::load(..).pixels(), raytracer-obj/src/color_map.rs:16-26; the `image`
// crate itself is a third-party dependency that is not part of the reference tree).
//   PNG: all colour types, bit depths 1-8, Adam7 interlace; inflate by zlib
//   BMP: uncompressed 8-bit palette / 24-bit / 32-bit
// 16-bit PNGs and compressed BMPs are rejected rather than guessed at.
#include "image_decode.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace rtimg {
namespace {

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
uint32_t le32(const uint8_t* p) { return ((uint32_t)p[3] << 24) | ((uint32_t)p[2] << 16) | ((uint32_t)p[1] << 8) | p[0]; }
uint32_t le16(const uint8_t* p) { return ((uint32_t)p[1] << 8) | p[0]; }

bool fail(std::string& err, const std::string& msg) { err = msg; return false; }

int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Reverses the scanline filters of one (sub)image in place; `rows` points at
// h * (1 + stride) bytes, each row led by its filter type.
bool unfilter(uint8_t* rows, uint32_t h, size_t stride, size_t bpp, std::string& err) {
    const uint8_t* prev = nullptr;
    for (uint32_t y = 0; y < h; y++) {
        uint8_t* row = rows + (size_t)y * (stride + 1);
        const uint8_t ft = row[0];
        uint8_t* cur = row + 1;
        if (ft > 4) return fail(err, "png: unknown filter type");
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev ? prev[i] : 0, c = (prev && i >= bpp) ? prev[i - bpp] : 0;
            int add = 0;
            if (ft == 1) add = a; else if (ft == 2) add = b; else if (ft == 3) add = (a + b) >> 1; else if (ft == 4) add = paeth(a, b, c);
            cur[i] = (uint8_t)(cur[i] + add);
        }
        prev = cur;
    }
    return true;
}

bool decode_png(const std::vector<uint8_t>& f, Image& out, std::string& err) {
    size_t p = 8;
    uint32_t w = 0, h = 0; int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    bool end = false;
    while (!end) {
        if (p + 12 > f.size()) return fail(err, "png: truncated");
        const uint32_t len = be32(&f[p]);
        if ((uint64_t)p + 12 + len > f.size()) return fail(err, "png: truncated chunk");
        const uint8_t* type = &f[p + 4]; const uint8_t* data = &f[p + 8];
        if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), type, len + 4) != be32(data + len)) return fail(err, "png: chunk CRC mismatch");
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13) return fail(err, "png: bad IHDR");
            w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
            if (data[10] != 0 || data[11] != 0 || interlace > 1) return fail(err, "png: unsupported compression/filter/interlace method");
        } else if (!memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!memcmp(type, "IEND", 4)) end = true;
        p += 12 + len;
    }
    int channels = 0;
    switch (ctype) { case 0: channels = 1; break; case 2: channels = 3; break; case 3: channels = 1; break; case 4: channels = 2; break; case 6: channels = 4; break;
                     default: return fail(err, "png: bad colour type"); }
    if (w == 0 || h == 0 || (uint64_t)w * h > (1ull << 28)) return fail(err, "png: bad dimensions");
    if (depth == 16) return fail(err, "png: 16-bit samples are not supported");
    if (!(depth == 8 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) return fail(err, "png: bad bit depth");
    if (ctype == 3 && plte.size() < 3) return fail(err, "png: palette missing");

    struct Pass { uint32_t x0, y0, dx, dy; };
    static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const Pass whole[1] = {{0, 0, 1, 1}};
    const Pass* passes = interlace ? adam7 : whole; const int n_pass = interlace ? 7 : 1;
    const size_t bits = (size_t)channels * depth, bpp = bits >= 8 ? bits / 8 : 1;
    size_t raw_size = 0;
    for (int k = 0; k < n_pass; k++) {
        const uint32_t pw = (w > passes[k].x0) ? (w - passes[k].x0 + passes[k].dx - 1) / passes[k].dx : 0;
        const uint32_t ph = (h > passes[k].y0) ? (h - passes[k].y0 + passes[k].dy - 1) / passes[k].dy : 0;
        if (pw && ph) raw_size += (size_t)ph * (1 + (pw * bits + 7) / 8);
    }
    std::vector<uint8_t> raw(raw_size);
    uLongf got = (uLongf)raw_size;
    const int zrc = uncompress(raw.data(), &got, idat.data(), (uLong)idat.size());
    if (zrc != Z_OK || got != raw_size) return fail(err, "png: inflate failed or size mismatch");

    out.width = w; out.height = h; out.rgba.assign((size_t)w * h * 4, 255);
    size_t off = 0;
    for (int k = 0; k < n_pass; k++) {
        const uint32_t pw = (w > passes[k].x0) ? (w - passes[k].x0 + passes[k].dx - 1) / passes[k].dx : 0;
        const uint32_t ph = (h > passes[k].y0) ? (h - passes[k].y0 + passes[k].dy - 1) / passes[k].dy : 0;
        if (!pw || !ph) continue;
        const size_t stride = (pw * bits + 7) / 8;
        if (!unfilter(raw.data() + off, ph, stride, bpp, err)) return false;
        for (uint32_t y = 0; y < ph; y++) {
            const uint8_t* row = raw.data() + off + (size_t)y * (stride + 1) + 1;
            for (uint32_t x = 0; x < pw; x++) {
                uint8_t s[4] = {0, 0, 0, 255};
                if (depth == 8) for (int c = 0; c < channels; c++) s[c] = row[(size_t)x * channels + c];
                else { const size_t bit = (size_t)x * depth; s[0] = (uint8_t)((row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1)); }
                uint8_t* px = &out.rgba[((size_t)(passes[k].y0 + y * passes[k].dy) * w + passes[k].x0 + x * passes[k].dx) * 4];
                if (ctype == 0) {
                    const uint8_t g = depth == 8 ? s[0] : (uint8_t)(s[0] * 255u / ((1u << depth) - 1));
                    px[0] = px[1] = px[2] = g;
                    px[3] = (trns.size() >= 2 && (uint32_t)((trns[0] << 8) | trns[1]) == s[0]) ? 0 : 255;
                } else if (ctype == 2) {
                    px[0] = s[0]; px[1] = s[1]; px[2] = s[2];
                    px[3] = (trns.size() >= 6 && trns[1] == s[0] && trns[3] == s[1] && trns[5] == s[2] && !trns[0] && !trns[2] && !trns[4]) ? 0 : 255;
                } else if (ctype == 3) {
                    if ((size_t)s[0] * 3 + 2 >= plte.size()) return fail(err, "png: palette index out of range");
                    px[0] = plte[s[0] * 3]; px[1] = plte[s[0] * 3 + 1]; px[2] = plte[s[0] * 3 + 2];
                    px[3] = s[0] < trns.size() ? trns[s[0]] : 255;
                } else if (ctype == 4) { px[0] = px[1] = px[2] = s[0]; px[3] = s[1]; }
                else { px[0] = s[0]; px[1] = s[1]; px[2] = s[2]; px[3] = s[3]; }
            }
        }
        off += (size_t)ph * (stride + 1);
    }
    return true;
}

bool decode_bmp(const std::vector<uint8_t>& f, Image& out, std::string& err) {
    if (f.size() < 54) return fail(err, "bmp: truncated header");
    const uint32_t data_off = le32(&f[10]), dib = le32(&f[14]);
    if (dib < 40) return fail(err, "bmp: unsupported header");
    const int32_t w = (int32_t)le32(&f[18]), hs = (int32_t)le32(&f[22]);
    const uint32_t bpp = le16(&f[28]), comp = le32(&f[30]);
    if (comp != 0) return fail(err, "bmp: compressed bitmaps are not supported");
    if (!(bpp == 8 || bpp == 24 || bpp == 32)) return fail(err, "bmp: unsupported bit count");
    const uint32_t h = (uint32_t)(hs < 0 ? -hs : hs);
    if (w <= 0 || h == 0 || (uint64_t)w * h > (1ull << 28)) return fail(err, "bmp: bad dimensions");
    const size_t stride = (((size_t)w * bpp + 31) / 32) * 4;
    if ((uint64_t)data_off + stride * h > f.size()) return fail(err, "bmp: truncated pixel data");
    uint32_t n_pal = le32(&f[46]); if (bpp == 8 && n_pal == 0) n_pal = 256;
    const uint8_t* pal = &f[14 + dib];
    if (bpp == 8 && (size_t)14 + dib + (size_t)n_pal * 4 > f.size()) return fail(err, "bmp: truncated palette");
    out.width = (uint32_t)w; out.height = h; out.rgba.assign((size_t)w * h * 4, 255);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t* row = &f[data_off + stride * (hs < 0 ? y : h - 1 - y)];      // bottom-up unless the height is negative
        for (int32_t x = 0; x < w; x++) {
            uint8_t* px = &out.rgba[((size_t)y * w + x) * 4];
            if (bpp == 8) { const uint32_t i = row[x]; if (i >= n_pal) return fail(err, "bmp: palette index out of range"); px[0] = pal[i * 4 + 2]; px[1] = pal[i * 4 + 1]; px[2] = pal[i * 4]; }
            else { const uint8_t* s = row + (size_t)x * (bpp / 8); px[0] = s[2]; px[1] = s[1]; px[2] = s[0]; }
        }
    }
    return true;
}

}  // namespace

bool decode_file(const std::string& path, Image& out, std::string& err) {
    FILE* fp = fopen(path.c_str(), "rb");
    if (!fp) return fail(err, "load_color_map: expected file at path " + path);          // builder.rs:161-163
    std::vector<uint8_t> f; uint8_t buf[65536]; size_t n;
    while ((n = fread(buf, 1, sizeof(buf), fp)) > 0) f.insert(f.end(), buf, buf + n);
    fclose(fp);
    static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    static const char lfs[] = "version https://git-lfs";
    if (f.size() >= 8 && !memcmp(f.data(), png_sig, 8)) return decode_png(f, out, err);
    if (f.size() >= 2 && f[0] == 'B' && f[1] == 'M') return decode_bmp(f, out, err);
    if (f.size() >= sizeof(lfs) - 1 && !memcmp(f.data(), lfs, sizeof(lfs) - 1)) { err = "git-lfs pointer, not an image: " + path; out.lfs_pointer = true; return false; }
    return fail(err, "load_color_map: Color map type unknown: " + path);                  // builder.rs:165-168
}

}  // namespace rtimg
