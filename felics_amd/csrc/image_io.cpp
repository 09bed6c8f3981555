// image_io.cpp -- see image_io.h.
#include "image_io.h"

#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <map>

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
    if (comp != 1) return "TIFF: compressed data is not supported (compression " + std::to_string(comp) + ")";
    if (bits != 8 && bits != 16) return "TIFF: " + std::to_string(bits) + " bits per sample are not supported";
    if (fmt != 1) return "TIFF: only unsigned integer samples are supported";
    if (planar != 1 && spp > 1) return "TIFF: planar configuration 2 is not supported";
    if (spp < 1 || spp > 4) return "TIFF: unsupported samples per pixel";
    if (photo > 2) return "TIFF: unsupported photometric interpretation " + std::to_string(photo);
    if (!tags.count(273)) return "TIFF: no strip offsets (tiled files are not supported)";
    const std::vector<uint32_t> &offs = tags[273];
    // RowsPerStrip defaults to "all rows" (2^32 - 1); zero is not a strip height: read it the same way
    uint32_t rps = first(278, H);
    if (rps == 0 || rps > H) rps = H;
    const size_t bps = bits / 8, row = (size_t)W * spp * bps;
    if ((uint64_t)row * H > (1ull << 34)) return "TIFF: image too large";
    // uncompressed samples live in the file: an image larger than the file is a forged header, refused before
    // anything is allocated for it
    if ((uint64_t)row * H > buf.size()) return "TIFF: the file is too short for the image it describes";
    const size_t nstrips = (H + rps - 1) / rps;
    if (offs.size() < nstrips) return "TIFF: strip table too short";
    img.width = W;
    img.height = H;
    img.channels = (int)spp;
    img.bits = (int)bits;
    img.data.resize(row * H);
    for (size_t s = 0; s < nstrips; s++) {
        const size_t rows = std::min<size_t>(rps, H - s * rps);
        if (!r.ok(offs[s], rows * row)) return "TIFF: strip data out of range";
        memcpy(img.data.data() + s * rps * row, buf.data() + offs[s], rows * row);
    }
    if (bits == 16 && r.le != host_is_little())
        for (size_t i = 0; i + 1 < img.data.size(); i += 2) std::swap(img.data[i], img.data[i + 1]);
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
    if (ext == "tif" || ext == "tiff") return read_tiff(buf, out);
    return "The image format could not be determined";
}

std::string write_image(const std::string &path, const Image &img) {
    const std::string ext = lower_ext(path);
    if (ext == "tif" || ext == "tiff") return write_tiff(path, img);
    if (ext == "pgm" || ext == "ppm" || ext == "pnm") return write_pnm(path, img);
    return "The image format could not be determined from the extension \"" + ext + "\" (supported: tiff, tif, pgm, ppm, pnm)";
}

}  // namespace imageio
