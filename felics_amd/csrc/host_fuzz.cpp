// host_fuzz -- sanitizer driver for the host-side parsers of untrusted input (CPU only, no GPU call that needs
// a device): every file of a directory goes through imageio::read_image (TIFF / PNM / PNG readers), through
// felics_read_header + felics_decompress (+ felics_decompress_with_header) with a bounded output buffer, and
// the argument checks of the encode entry points are exercised with bad arguments.  Built with
// -fsanitize=address,undefined by `make asan`; tests/test_host_hardening.py feeds it a mutated corpus.
// Exit code 0 = every input was handled (accepted or rejected) without a sanitizer report.
#include <dirent.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/felics.h"
#include "image_io.h"

static bool slurp(const std::string &p, std::vector<uint8_t> &b) {
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) return false;
    uint8_t chunk[1 << 16];
    size_t got;
    b.clear();
    while ((got = fread(chunk, 1, sizeof chunk, f)) > 0) b.insert(b.end(), chunk, chunk + got);
    fclose(f);
    return true;
}

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: host_fuzz DIR\n");
        return 2;
    }
    // argument checks that must hold without any device
    {
        size_t n = 0;
        uint8_t out[32];
        felics_header h{0, 0, 1, 1};
        int bad = 0;
        bad += felics_compress(nullptr, out, 1, 1, 0, 0, out, sizeof out, &n) == FELICS_OK;
        bad += felics_compress_batch(nullptr, 1, nullptr, 1, 1, 0, 0, nullptr, nullptr, nullptr) == FELICS_OK;
        bad += felics_write_header(&h, out, 3) != FELICS_E_BUFFER_TOO_SMALL;
        h.color_type = 7;
        bad += felics_write_header(&h, out, sizeof out) != FELICS_E_INVALID_COLOR_TYPE;
        bad += felics_read_header(nullptr, 0, &h) != FELICS_E_IO;
        bad += felics_decompress_with_header(out, 0, nullptr, out, sizeof out) != FELICS_E_INVALID_ARGUMENT;
        bad += felics_get_stats(nullptr, nullptr) != FELICS_E_INVALID_ARGUMENT;
        felics_ctx *ctx = nullptr;
        bad += felics_ctx_create(-1, &ctx) == FELICS_OK;
        if (bad) {
            fprintf(stderr, "argument checks: %d unexpected results\n", bad);
            return 1;
        }
    }
    DIR *d = opendir(argv[1]);
    if (!d) return 2;
    std::vector<std::string> names;
    while (dirent *e = readdir(d))
        if (e->d_name[0] != '.') names.push_back(e->d_name);
    closedir(d);
    size_t accepted = 0, rejected = 0;
    std::vector<uint8_t> buf, px(64u << 20);  // decoders get a bounded buffer whatever the header claims
    for (const std::string &n : names) {
        const std::string path = std::string(argv[1]) + "/" + n;
        imageio::Image img;
        bool open_failed = false;
        const std::string err = imageio::read_image(path, img, open_failed);
        (err.empty() ? accepted : rejected)++;
        if (!slurp(path, buf)) continue;
        felics_header h;
        const int rc = felics_decompress(buf.data(), buf.size(), px.data(), px.size(), &h);
        (rc == FELICS_OK ? accepted : rejected)++;
        if (felics_read_header(buf.data(), buf.size(), &h) == FELICS_OK) {
            const int rc2 = felics_decompress_with_header(buf.data() + FELICS_HEADER_BYTES, buf.size() - FELICS_HEADER_BYTES,
                                                          &h, px.data(), px.size());
            if ((rc2 == FELICS_OK) != (rc == FELICS_OK)) {
                fprintf(stderr, "%s: decompress %d but decompress_with_header %d\n", n.c_str(), rc, rc2);
                return 1;
            }
        }
    }
    printf("host_fuzz: %zu files, %zu accepted, %zu rejected\n", names.size(), accepted, rejected);
    return 0;
}
