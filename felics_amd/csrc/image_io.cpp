// image_io.cpp -- see image_io.h.
#include "image_io.h"

#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <map>

#include <zlib.h>

namespace imageio {

std::string Image::color_name() const {
    const char *base = channels == 1 ? "L" : channels == 2 ? "La" : channels == 3 ? "Rgb" : "Rgba";
    return std::string(base) + std::to_string(bits);
}

namespace {

bool host_is_little() {
    const uint16_t v = 1;
    return *reinterpret_cast<const uint8_t *>(&v) == 1;
}

std::string lower_ext(const std::string &path) {
    size_t dot = path.find_last_of('.');
    size_t slash = path.find_last_of("/\\");
    if (dot == std::string::npos || (slash != std::string::npos && dot < slash)) return "";
    std::string e = path.substr(dot + 1);
    std::transform(e.begin(), e.end(), e.begin(), [](unsigned char c) { return (char)std::tolower(c); });
    return e;
}

bool slurp(const std::string &path, std::vector<uint8_t> &buf, std::string &err) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) {
        err = strerror(errno);
        return false;
    }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf.resize(n > 0 ? (size_t)n : 0);
    size_t got = buf.empty() ? 0 : fread(buf.data(), 1, buf.size(), f);
    fclose(f);
    if (got != buf.size()) {
        err = "short read";
        return false;
    }
    return true;
}

// ---------------- TIFF ----------------

struct TiffReader {
    const std::vector<uint8_t> &b;
    bool le = true;
    explicit TiffReader(const std::vector<uint8_t> &buf) : b(buf) {}
    bool ok(size_t off, size_t n) const { return off <= b.size() && n <= b.size() - off; }
    uint16_t u16(size_t o) const { return le ? (uint16_t)(b[o] | (b[o + 1] << 8)) : (uint16_t)((b[o] << 8) | b[o + 1]); }
    uint32_t u32(size_t o) const {
        return le ? ((uint32_t)b[o] | ((uint32_t)b[o + 1] << 8) | ((uint32_t)b[o + 2] << 16) | ((uint32_t)b[o + 3] << 24))
                  : (((uint32_t)b[o] << 24) | ((uint32_t)b[o + 1] << 16) | ((uint32_t)b[o + 2] << 8) | (uint32_t)b[o + 3]);
    }
};


// ---------------- decompressors shared by the TIFF and PNG readers ----------------

// zlib stream -> exactly `want` bytes (more or fewer is an error)
std::string inflate_exact(const uint8_t *src, size_t n, uint8_t *dst, size_t want) {
    z_stream z;
    memset(&z, 0, sizeof z);
    if (inflateInit(&z) != Z_OK) return "zlib: cannot start";
    z.next_in = const_cast<uint8_t *>(src);
    z.next_out = dst;
    size_t in_left = n, out_left = want;
    int rc = Z_OK;
    uint8_t spill[64];
    while (rc == Z_OK) {
        z.avail_in = (uInt)std::min<size_t>(in_left, 1u << 30);
        in_left -= z.avail_in;
        const bool spilling = out_left == 0;
        if (spilling) {
            z.next_out = spill;
            z.avail_out = sizeof spill;
        } else {
            z.avail_out = (uInt)std::min<size_t>(out_left, 1u << 30);
            out_left -= z.avail_out;
        }
        rc = inflate(&z, Z_NO_FLUSH);
        in_left += z.avail_in;
        if (spilling) {
            if (z.avail_out != sizeof spill) {
                inflateEnd(&z);
                return "compressed data holds more than the image needs";
            }
        } else {
            out_left += z.avail_out;
        }
        if (rc == Z_BUF_ERROR || (rc == Z_OK && in_left == 0 && z.avail_in == 0 && z.avail_out != 0)) break;
    }
    inflateEnd(&z);
    if (rc != Z_STREAM_END && !(rc == Z_OK && out_left == 0)) return "corrupt deflate stream";
    if (out_left != 0) return "compressed data ends before the image is complete";
    return "";
}

// PackBits (TIFF 6.0 section 9)
std::string packbits(const uint8_t *src, size_t n, uint8_t *dst, size_t want) {
    size_t i = 0, o = 0;
    while (o < want) {
        if (i >= n) return "PackBits data ends early";
        const int8_t c = (int8_t)src[i++];
        if (c >= 0) {
            const size_t len = (size_t)c + 1;
            if (len > n - i || len > want - o) return "PackBits run out of range";
            memcpy(dst + o, src + i, len);
            i += len;
            o += len;
        } else if (c != -128) {
            const size_t len = (size_t)(-c) + 1;
            if (i >= n || len > want - o) return "PackBits run out of range";
            memset(dst + o, src[i++], len);
            o += len;
        }
    }
    return "";
}

// TIFF LZW (TIFF 6.0 section 13): MSB-first codes of 9..12 bits, ClearCode 256, EndOfInformation 257, the code
// width grows one code early
std::string tiff_lzw(const uint8_t *src, size_t n, uint8_t *dst, size_t want) {
    std::vector<uint16_t> prefix(4096);
    std::vector<uint8_t> suffix(4096), first_byte(4096);
    std::vector<uint16_t> length(4096);
    for (int i = 0; i < 256; i++) {
        suffix[i] = first_byte[i] = (uint8_t)i;
        length[i] = 1;
    }
    uint32_t acc = 0, nacc = 0;
    size_t i = 0, o = 0;
    unsigned width = 9, next = 258;
    int prev = -1;
    while (o < want) {
        while (nacc < width) {
            if (i >= n) return "LZW data ends early";
            acc = (acc << 8) | src[i++];
            nacc += 8;
        }
        const unsigned code = (acc >> (nacc - width)) & ((1u << width) - 1u);
        nacc -= width;
        if (code == 257) break;
        if (code == 256) {
            width = 9;
            next = 258;
            prev = -1;
            continue;
        }
        unsigned entry = code;
        if (prev < 0) {
            if (code >= 256) return "corrupt LZW stream";
        } else if (code > next || (code == next && next >= 4096)) {
            return "corrupt LZW stream";
        } else {
            if (next < 4096) {  // new entry: string(prev) + first byte of string(code) (of string(prev) when code is the new entry)
                prefix[next] = (uint16_t)prev;
                first_byte[next] = first_byte[prev];
                length[next] = (uint16_t)(length[prev] + 1);
                suffix[next] = code == next ? first_byte[prev] : first_byte[code];
                next++;
            }
        }
        const size_t len = length[entry];
        if (len > want - o) {  // the last code may run past the strip: keep what fits
            std::vector<uint8_t> tmp(len);
            unsigned c = entry;
            for (size_t k = len; k-- > 0;) {
                tmp[k] = suffix[c];
                c = prefix[c];
            }
            memcpy(dst + o, tmp.data(), want - o);
            o = want;
            break;
        }
        unsigned c = entry;
        for (size_t k = len; k-- > 0;) {
            dst[o + k] = suffix[c];
            c = prefix[c];
        }
        o += len;
        prev = (int)entry;
        if (next + 1 >= (1u << width) && width < 12) width++;  // "early change"
    }
    if (o != want) return "LZW data ends before the strip is complete";
    return "";
}

// ---------------- PNG (ISO/IEC 15948): what `image` decodes for cfelics (src/bin/cfelics.rs:36-44) ----------------

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return pa <= pb && pa <= pc ? a : (pb <= pc ? b : c);
}

// undo the filters of `rows` scanlines of `rowbytes` bytes each (a filter byte in front of every line), in place;
// bpp = bytes per complete pixel, at least 1
std::string png_unfilter(uint8_t *data, size_t rows, size_t rowbytes, size_t bpp) {
    std::vector<uint8_t> zero(rowbytes, 0);
    const uint8_t *prev = zero.data();
    for (size_t y = 0; y < rows; y++) {
        uint8_t *line = data + y * (rowbytes + 1);
        const uint8_t ft = line[0];
        uint8_t *cur = line + 1;
        if (ft > 4) return "PNG: unknown filter type";
        for (size_t i = 0; i < rowbytes; i++) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            const int add = ft == 0 ? 0 : ft == 1 ? a : ft == 2 ? b : ft == 3 ? (a + b) / 2 : paeth(a, b, c);
            cur[i] = (uint8_t)(cur[i] + add);
        }
        prev = cur;
    }
    return "";
}

std::string read_png(const std::vector<uint8_t> &buf, Image &img) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    if (buf.size() < 8 || memcmp(buf.data(), sig, 8) != 0) return "PNG signature not found";
    size_t pos = 8;
    uint32_t W = 0, H = 0;
    unsigned depth = 0, ctype = 0, interlace = 0;
    bool have_ihdr = false, have_trns = false, done = false;
    std::vector<uint8_t> idat, palette;
    while (!done) {
        if (buf.size() - pos < 12) return "PNG: truncated chunk";
        const uint32_t len = be32(&buf[pos]);
        if (len > buf.size() - pos - 12) return "PNG: truncated chunk";
        const uint8_t *type = &buf[pos + 4], *body = &buf[pos + 8];
        if ((uint32_t)crc32(crc32(0, nullptr, 0), type, len + 4) != be32(body + len)) return "PNG: CRC mismatch";
        if (memcmp(type, "IHDR", 4) == 0) {
            if (len != 13 || have_ihdr) return "PNG: bad IHDR";
            W = be32(body);
            H = be32(body + 4);
            depth = body[8];
            ctype = body[9];
            interlace = body[12];
            if (body[10] != 0 || body[11] != 0 || interlace > 1) return "PNG: unknown compression, filter or interlace method";
            have_ihdr = true;
        } else if (!have_ihdr) {
            return "PNG: IHDR is not the first chunk";
        } else if (memcmp(type, "PLTE", 4) == 0) {
            if (len % 3 != 0 || len > 768) return "PNG: bad PLTE";
            palette.assign(body, body + len);
        } else if (memcmp(type, "tRNS", 4) == 0) {
            have_trns = true;
        } else if (memcmp(type, "IDAT", 4) == 0) {
            idat.insert(idat.end(), body, body + len);
        } else if (memcmp(type, "IEND", 4) == 0) {
            done = true;
        } else if (!(type[0] & 0x20)) {
            return "PNG: unknown critical chunk";
        }
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || W == 0 || H == 0) return "PNG: missing image dimensions";
    const unsigned chans = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    const bool depth_ok = ctype == 0 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                          : ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8)
                                       : (depth == 8 || depth == 16);
    if (chans == 0 || !depth_ok) return "PNG: invalid colour type / bit depth";
    if (ctype == 3 && palette.empty()) return "PNG: palette image without PLTE";
    const size_t bits_pp = (size_t)chans * depth, bpp = std::max<size_t>(1, bits_pp / 8);
    if ((uint64_t)W * H * std::max<size_t>(bits_pp, 8) / 8 > (1ull << 34)) return "PNG: image too large";
    // the passes: one for a plain image, seven for Adam7
    struct Pass { uint32_t x0, y0, dx, dy; };
    static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const Pass whole = {0, 0, 1, 1};
    const Pass *passes = interlace ? adam7 : &whole;
    const int npass = interlace ? 7 : 1;
    size_t total = 0;
    for (int p = 0; p < npass; p++) {
        const uint64_t pw = (W + passes[p].dx - 1 - passes[p].x0) / passes[p].dx, ph = (H + passes[p].dy - 1 - passes[p].y0) / passes[p].dy;
        if (W <= passes[p].x0 || H <= passes[p].y0 || pw == 0 || ph == 0) continue;
        total += (size_t)ph * (1 + (size_t)((pw * bits_pp + 7) / 8));
    }
    // deflate cannot expand a byte into more than ~1030: a header that needs more than the file can give is forged
    if (total / 1032 > idat.size() + 1) return "PNG: the file is too short for the image it describes";
    std::vector<uint8_t> raw(total);
    std::string e = inflate_exact(idat.data(), idat.size(), raw.data(), raw.size());
    if (!e.empty()) return "PNG: " + e;
    // output layout
    const bool to16 = depth == 16;
    const unsigned out_ch = ctype == 3 ? (have_trns ? 4 : 3) : chans;
    img.width = W;
    img.height = H;
    img.channels = (int)out_ch;
    img.bits = to16 ? 16 : 8;
    img.data.assign((size_t)W * H * out_ch * (to16 ? 2 : 1), ctype == 3 && have_trns ? 255 : 0);
    size_t at = 0;
    for (int p = 0; p < npass; p++) {
        if (W <= passes[p].x0 || H <= passes[p].y0) continue;
        const uint32_t pw = (W + passes[p].dx - 1 - passes[p].x0) / passes[p].dx, ph = (H + passes[p].dy - 1 - passes[p].y0) / passes[p].dy;
        if (pw == 0 || ph == 0) continue;
        const size_t rowbytes = ((size_t)pw * bits_pp + 7) / 8;
        e = png_unfilter(raw.data() + at, ph, rowbytes, bpp);
        if (!e.empty()) return e;
        for (uint32_t py = 0; py < ph; py++) {
            const uint8_t *line = raw.data() + at + (size_t)py * (rowbytes + 1) + 1;
            const uint32_t y = passes[p].y0 + py * passes[p].dy;
            for (uint32_t px = 0; px < pw; px++) {
                const uint32_t x = passes[p].x0 + px * passes[p].dx;
                uint8_t *dst = img.data.data() + ((size_t)y * W + x) * out_ch * (to16 ? 2 : 1);
                if (depth == 16) {
                    for (unsigned c = 0; c < chans; c++) {
                        const uint16_t v = (uint16_t)((line[(px * chans + c) * 2] << 8) | line[(px * chans + c) * 2 + 1]);
                        memcpy(dst + c * 2, &v, 2);
                    }
                } else if (depth == 8) {
                    if (ctype == 3) {
                        const unsigned idx = line[px];
                        if ((size_t)idx * 3 + 3 > palette.size()) return "PNG: palette index out of range";
                        memcpy(dst, &palette[idx * 3], 3);
                    } else {
                        memcpy(dst, line + (size_t)px * chans, chans);
                    }
                } else {  // 1, 2 or 4 bits: gray is scaled to 8 bits, palette indices are looked up
                    const unsigned per = 8 / depth, sh = (per - 1 - px % per) * depth;
                    const unsigned v = (line[px / per] >> sh) & ((1u << depth) - 1u);
                    if (ctype == 3) {
                        if ((size_t)v * 3 + 3 > palette.size()) return "PNG: palette index out of range";
                        memcpy(dst, &palette[v * 3], 3);
                    } else {
                        dst[0] = (uint8_t)(v * 255u / ((1u << depth) - 1u));
                    }
                }
            }
        }
        at += (size_t)ph * (rowbytes + 1);
    }
    return "";
}

void png_chunk(std::vector<uint8_t> &o, const char *type, const uint8_t *body, size_t len) {
    const size_t at = o.size();
    o.resize(at + 12 + len);
    o[at] = (uint8_t)(len >> 24); o[at + 1] = (uint8_t)(len >> 16); o[at + 2] = (uint8_t)(len >> 8); o[at + 3] = (uint8_t)len;
    memcpy(&o[at + 4], type, 4);
    if (len) memcpy(&o[at + 8], body, len);
    const uint32_t crc = (uint32_t)crc32(crc32(0, nullptr, 0), &o[at + 4], (uInt)(len + 4));
    o[at + 8 + len] = (uint8_t)(crc >> 24); o[at + 9 + len] = (uint8_t)(crc >> 16); o[at + 10 + len] = (uint8_t)(crc >> 8); o[at + 11 + len] = (uint8_t)crc;
}

std::string write_png(const std::string &path, const Image &img) {
    if (img.channels < 1 || img.channels > 4 || (img.bits != 8 && img.bits != 16)) return "PNG: unsupported sample layout";
    const size_t bps = img.bits / 8, row = (size_t)img.width * img.channels * bps;
    std::vector<uint8_t> raw((row + 1) * img.height);
    for (uint32_t y = 0; y < img.height; y++) {
        uint8_t *line = &raw[(size_t)y * (row + 1)];
        line[0] = 0;  // filter type None
        const uint8_t *src = img.data.data() + (size_t)y * row;
        if (img.bits == 16 && host_is_little())
            for (size_t i = 0; i + 1 < row; i += 2) {
                line[1 + i] = src[i + 1];
                line[2 + i] = src[i];
            }
        else if (row)
            memcpy(line + 1, src, row);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return "PNG: deflate failed";
    std::vector<uint8_t> o = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    uint8_t ihdr[13];
    for (int i = 0; i < 4; i++) {
        ihdr[i] = (uint8_t)(img.width >> (24 - 8 * i));
        ihdr[4 + i] = (uint8_t)(img.height >> (24 - 8 * i));
    }
    ihdr[8] = (uint8_t)img.bits;
    ihdr[9] = img.channels == 1 ? 0 : img.channels == 2 ? 4 : img.channels == 3 ? 2 : 6;
    ihdr[10] = ihdr[11] = ihdr[12] = 0;
    png_chunk(o, "IHDR", ihdr, 13);
    png_chunk(o, "IDAT", comp.data(), clen);
    png_chunk(o, "IEND", nullptr, 0);
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return strerror(errno);
    const bool ok = fwrite(o.data(), 1, o.size(), f) == o.size();
    if (fclose(f) != 0 || !ok) return "write failed";
    return "";
}

std::string read_tiff(const std::vector<uint8_t> &buf, Image &img) {
    TiffReader r(buf);
    if (buf.size() < 8) return "TIFF: file too short";
    if (buf[0] == 'I' && buf[1] == 'I')
        r.le = true;
    else if (buf[0] == 'M' && buf[1] == 'M')
        r.le = false;
    else
        return "TIFF signature not found";
    if (r.u16(2) != 42) return "TIFF signature invalid (BigTIFF is not supported)";
    size_t ifd = r.u32(4);
    if (!r.ok(ifd, 2)) return "TIFF: bad IFD offset";
    const unsigned nent = r.u16(ifd);
    if (!r.ok(ifd + 2, (size_t)nent * 12)) return "TIFF: truncated IFD";
    std::map<uint16_t, std::vector<uint32_t>> tags;
    for (unsigned i = 0; i < nent; i++) {
        const size_t e = ifd + 2 + (size_t)i * 12;
        const uint16_t tag = r.u16(e), type = r.u16(e + 2);
        const uint32_t count = r.u32(e + 4);
        size_t tsz = type == 1 || type == 2 || type == 6 || type == 7 ? 1 : type == 3 || type == 8 ? 2 : type == 4 || type == 9 ? 4 : 0;
        if (tsz == 0) continue;  // rationals etc.: not needed
        if (count > (1u << 26)) return "TIFF: unreasonable tag count";
        size_t off = (size_t)count * tsz <= 4 ? e + 8 : r.u32(e + 8);
        if (!r.ok(off, (size_t)count * tsz)) return "TIFF: tag data out of range";
        std::vector<uint32_t> v(count);
        for (uint32_t j = 0; j < count; j++) v[j] = tsz == 1 ? buf[off + j] : tsz == 2 ? r.u16(off + 2 * (size_t)j) : r.u32(off + 4 * (size_t)j);
        tags[tag] = std::move(v);
    }
    auto first = [&](uint16_t tag, uint32_t dflt) { auto it = tags.find(tag); return it == tags.end() || it->second.empty() ? dflt : it->second[0]; };
    const uint32_t W = first(256, 0), H = first(257, 0);
    const uint32_t spp = first(277, 1), comp = first(259, 1), photo = first(262, 1), planar = first(284, 1);
    const uint32_t fmt = first(339, 1);
    uint32_t bits = first(258, 1);
    if (tags.count(258))
        for (uint32_t v : tags[258])
            if (v != bits) return "TIFF: mixed bits per sample are not supported";
    if (W == 0 || H == 0) return "TIFF: missing image dimensions";
    if (comp != 1 && comp != 5 && comp != 8 && comp != 32946 && comp != 32773)
        return "TIFF: compression " + std::to_string(comp) + " is not supported (none, LZW, Deflate and PackBits are)";
    if (bits != 8 && bits != 16) return "TIFF: " + std::to_string(bits) + " bits per sample are not supported";
    if (fmt != 1) return "TIFF: only unsigned integer samples are supported";
    if (planar != 1 && spp > 1) return "TIFF: planar configuration 2 is not supported";
    if (spp < 1 || spp > 4) return "TIFF: unsupported samples per pixel";
    if (photo > 2) return "TIFF: unsupported photometric interpretation " + std::to_string(photo);
    if (!tags.count(273)) return "TIFF: no strip offsets (tiled files are not supported)";
    const uint32_t predictor = first(317, 1);
    if (predictor != 1 && predictor != 2) return "TIFF: predictor " + std::to_string(predictor) + " is not supported";
    const std::vector<uint32_t> &offs = tags[273];
    // RowsPerStrip defaults to "all rows" (2^32 - 1); zero is not a strip height: read it the same way
    uint32_t rps = first(278, H);
    if (rps == 0 || rps > H) rps = H;
    const size_t bps = bits / 8, row = (size_t)W * spp * bps;
    if ((uint64_t)row * H > (1ull << 34)) return "TIFF: image too large";
    // A forged header must not size an allocation: uncompressed samples live in the file, and none of the
    // supported compressions expands a byte into more than ~1030 (Deflate's bound; LZW and PackBits stay far below).
    if (comp == 1 ? (uint64_t)row * H > buf.size() : (uint64_t)row * H / 1032 > buf.size())
        return "TIFF: the file is too short for the image it describes";
    const size_t nstrips = (H + rps - 1) / rps;
    if (offs.size() < nstrips) return "TIFF: strip table too short";
    if (comp != 1 && (!tags.count(279) || tags[279].size() < nstrips)) return "TIFF: compressed strips need StripByteCounts";
    img.width = W;
    img.height = H;
    img.channels = (int)spp;
    img.bits = (int)bits;
    img.data.resize(row * H);
    for (size_t s = 0; s < nstrips; s++) {
        const size_t rows = std::min<size_t>(rps, H - s * rps);
        uint8_t *dst = img.data.data() + s * rps * row;
        if (comp == 1) {
            if (!r.ok(offs[s], rows * row)) return "TIFF: strip data out of range";
            memcpy(dst, buf.data() + offs[s], rows * row);
            continue;
        }
        const size_t clen = tags[279][s];
        if (!r.ok(offs[s], clen)) return "TIFF: strip data out of range";
        const std::string e = comp == 5         ? tiff_lzw(buf.data() + offs[s], clen, dst, rows * row)
                              : comp == 32773   ? packbits(buf.data() + offs[s], clen, dst, rows * row)
                                                : inflate_exact(buf.data() + offs[s], clen, dst, rows * row);
        if (!e.empty()) return "TIFF: " + e;
    }
    if (bits == 16 && r.le != host_is_little())
        for (size_t i = 0; i + 1 < img.data.size(); i += 2) std::swap(img.data[i], img.data[i + 1]);
    if (predictor == 2) {  // horizontal differencing, per sample (TIFF 6.0 section 14), undone on native-endian samples
        for (uint32_t y = 0; y < H; y++) {
            if (bits == 8) {
                uint8_t *p = img.data.data() + (size_t)y * row;
                for (size_t i = spp; i < (size_t)W * spp; i++) p[i] = (uint8_t)(p[i] + p[i - spp]);
            } else {
                uint16_t *p = reinterpret_cast<uint16_t *>(img.data.data() + (size_t)y * row);
                for (size_t i = spp; i < (size_t)W * spp; i++) p[i] = (uint16_t)(p[i] + p[i - spp]);
            }
        }
    }
    if (photo == 0) {  // WhiteIsZero -> store as BlackIsZero
        if (bits == 8)
            for (auto &v : img.data) v = (uint8_t)(255 - v);
        else
            for (size_t i = 0; i + 1 < img.data.size(); i += 2) {
                uint16_t v;
                memcpy(&v, &img.data[i], 2);
                v = (uint16_t)(65535 - v);
                memcpy(&img.data[i], &v, 2);
            }
    }
    return "";
}

void put16(std::vector<uint8_t> &o, uint16_t v) {
    o.push_back((uint8_t)v);
    o.push_back((uint8_t)(v >> 8));
}
void put32(std::vector<uint8_t> &o, uint32_t v) {
    for (int i = 0; i < 4; i++) o.push_back((uint8_t)(v >> (8 * i)));
}

std::string write_tiff(const std::string &path, const Image &img) {
    // little-endian baseline TIFF, one strip, pixel data right after the 8-byte header
    const size_t bps = img.bits / 8, nbytes = (size_t)img.width * img.height * img.channels * bps;
    if (nbytes + 1024 > 0xFFFFFFFFull) return "image too large for a classic TIFF";
    std::vector<uint8_t> o;
    o.reserve(nbytes + 256);
    o.push_back('I');
    o.push_back('I');
    put16(o, 42);
    const uint32_t ifd_off = (uint32_t)(8 + nbytes + (nbytes & 1));
    put32(o, ifd_off);
    const size_t data_at = o.size();
    o.resize(o.size() + nbytes);
    if (nbytes) memcpy(&o[data_at], img.data.data(), nbytes);
    if (img.bits == 16 && !host_is_little())
        for (size_t i = data_at; i + 1 < data_at + nbytes; i += 2) std::swap(o[i], o[i + 1]);
    if (nbytes & 1) o.push_back(0);
    struct Ent { uint16_t tag, type; uint32_t count, value; };
    std::vector<Ent> ents;
    const uint32_t bits_off = ifd_off + 2 + 10 * 12 + 4;  // BitsPerSample array for RGB lives after the IFD
    ents.push_back({256, 4, 1, img.width});
    ents.push_back({257, 4, 1, img.height});
    if (img.channels == 1)
        ents.push_back({258, 3, 1, (uint32_t)img.bits});
    else
        ents.push_back({258, 3, (uint32_t)img.channels, bits_off});
    ents.push_back({259, 3, 1, 1});
    ents.push_back({262, 3, 1, img.channels >= 3 ? 2u : 1u});
    ents.push_back({273, 4, 1, 8});
    ents.push_back({277, 3, 1, (uint32_t)img.channels});
    ents.push_back({278, 4, 1, img.height});
    ents.push_back({279, 4, 1, (uint32_t)nbytes});
    ents.push_back({284, 3, 1, 1});
    put16(o, (uint16_t)ents.size());
    for (const Ent &e : ents) {
        put16(o, e.tag);
        put16(o, e.type);
        put32(o, e.count);
        if (e.type == 3 && e.count == 1) {
            put16(o, (uint16_t)e.value);
            put16(o, 0);
        } else {
            put32(o, e.value);
        }
    }
    put32(o, 0);
    if (img.channels > 1)
        for (int c = 0; c < img.channels; c++) put16(o, (uint16_t)img.bits);
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return strerror(errno);
    const bool ok = fwrite(o.data(), 1, o.size(), f) == o.size();
    if (fclose(f) != 0 || !ok) return "write failed";
    return "";
}

// ---------------- PNM ----------------

bool pnm_token(const std::vector<uint8_t> &b, size_t &pos, uint32_t &val) {
    for (;;) {
        while (pos < b.size() && std::isspace(b[pos])) pos++;
        if (pos < b.size() && b[pos] == '#') {
            while (pos < b.size() && b[pos] != '\n') pos++;
            continue;
        }
        break;
    }
    if (pos >= b.size() || !std::isdigit(b[pos])) return false;
    uint64_t v = 0;
    while (pos < b.size() && std::isdigit(b[pos])) {
        v = v * 10 + (b[pos++] - '0');
        if (v > 0xFFFFFFFFull) return false;
    }
    val = (uint32_t)v;
    return true;
}

std::string read_pnm(const std::vector<uint8_t> &buf, Image &img) {
    if (buf.size() < 3 || buf[0] != 'P' || (buf[1] != '5' && buf[1] != '6')) return "PNM: only binary P5/P6 are supported";
    size_t pos = 2;
    uint32_t W, H, maxv;
    if (!pnm_token(buf, pos, W) || !pnm_token(buf, pos, H) || !pnm_token(buf, pos, maxv)) return "PNM: bad header";
    if (pos >= buf.size() || !std::isspace(buf[pos])) return "PNM: bad header";
    pos++;
    if (maxv != 255 && maxv != 65535) return "PNM: maxval must be 255 or 65535";
    img.width = W;
    img.height = H;
    img.channels = buf[1] == '5' ? 1 : 3;
    img.bits = maxv == 255 ? 8 : 16;
    const size_t nbytes = (size_t)W * H * img.channels * (img.bits / 8);
    if (buf.size() - pos < nbytes) return "PNM: truncated pixel data";
    img.data.assign(buf.begin() + pos, buf.begin() + pos + nbytes);
    if (img.bits == 16 && host_is_little())  // PNM is big-endian
        for (size_t i = 0; i + 1 < nbytes; i += 2) std::swap(img.data[i], img.data[i + 1]);
    return "";
}

std::string write_pnm(const std::string &path, const Image &img) {
    if (img.channels != 1 && img.channels != 3) return "PNM holds gray or RGB only";
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return strerror(errno);
    fprintf(f, "P%c\n%u %u\n%u\n", img.channels == 1 ? '5' : '6', img.width, img.height, img.bits == 8 ? 255u : 65535u);
    std::vector<uint8_t> tmp;
    const uint8_t *p = img.data.data();
    if (img.bits == 16 && host_is_little()) {
        tmp = img.data;
        for (size_t i = 0; i + 1 < tmp.size(); i += 2) std::swap(tmp[i], tmp[i + 1]);
        p = tmp.data();
    }
    const bool ok = img.data.empty() || fwrite(p, 1, img.data.size(), f) == img.data.size();
    if (fclose(f) != 0 || !ok) return "write failed";
    return "";
}

}  // namespace

std::string read_image(const std::string &path, Image &out, bool &open_failed) {
    open_failed = false;
    std::vector<uint8_t> buf;
    std::string err;
    if (!slurp(path, buf, err)) {
        open_failed = true;
        return err;
    }
    const std::string ext = lower_ext(path);
    if (buf.size() >= 2 && ((buf[0] == 'I' && buf[1] == 'I') || (buf[0] == 'M' && buf[1] == 'M'))) return read_tiff(buf, out);
    if (buf.size() >= 2 && buf[0] == 'P' && (buf[1] == '5' || buf[1] == '6')) return read_pnm(buf, out);
    if (buf.size() >= 4 && buf[0] == 0x89 && buf[1] == 'P' && buf[2] == 'N' && buf[3] == 'G') return read_png(buf, out);
    if (ext == "tif" || ext == "tiff") return read_tiff(buf, out);
    if (ext == "png") return read_png(buf, out);
    return "The image format could not be determined";
}

std::string write_image(const std::string &path, const Image &img) {
    const std::string ext = lower_ext(path);
    if (ext == "tif" || ext == "tiff") return write_tiff(path, img);
    if (ext == "pgm" || ext == "ppm" || ext == "pnm") return write_pnm(path, img);
    if (ext == "png") return write_png(path, img);
    return "The image format could not be determined from the extension \"" + ext + "\" (supported: tiff, tif, png, pgm, ppm, pnm)";
}

}  // namespace imageio
