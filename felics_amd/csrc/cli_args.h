// cli_args.h -- the two-flag command line both tools share (clap derive in the reference:
// src/bin/cfelics.rs:11-22, src/bin/dfelics.rs:9-21): -i/--input <PATH>, -o/--output <PATH>,
// -h/--help, -V/--version; usage errors exit with status 2 like clap.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

struct CliArgs {
    std::string input, output;
    int device = 0;
};

inline void cli_usage(FILE *f, const char *prog, const char *about, const char *in_help, const char *out_help) {
    fprintf(f, "%s\n\nUsage: %s --input <INPUT> --output <OUTPUT>\n\nOptions:\n"
               "  -i, --input <INPUT>    %s\n  -o, --output <OUTPUT>  %s\n"
               "      --device <N>       HIP device to encode on [default: 0]\n"
               "  -h, --help             Print help\n  -V, --version          Print version\n",
            about, prog, in_help, out_help);
}

inline CliArgs cli_parse(int argc, char **argv, const char *prog, const char *about, const char *in_help,
                         const char *out_help) {
    CliArgs a;
    bool have_in = false, have_out = false;
    for (int i = 1; i < argc; i++) {
        std::string s = argv[i];
        auto value = [&](const std::string &flag) -> std::string {
            size_t eq = s.find('=');
            if (s.rfind("--", 0) == 0 && eq != std::string::npos) return s.substr(eq + 1);
            if (i + 1 >= argc) {
                fprintf(stderr, "error: a value is required for '%s <VALUE>' but none was supplied\n", flag.c_str());
                exit(2);
            }
            return argv[++i];
        };
        const std::string name = s.rfind("--", 0) == 0 ? s.substr(0, s.find('=')) : s;
        if (name == "-h" || name == "--help") {
            cli_usage(stdout, prog, about, in_help, out_help);
            exit(0);
        } else if (name == "-V" || name == "--version") {
            printf("%s 0.1.0\n", prog);
            exit(0);
        } else if (name == "-i" || name == "--input") {
            a.input = value("--input");
            have_in = true;
        } else if (name == "-o" || name == "--output") {
            a.output = value("--output");
            have_out = true;
        } else if (name == "--device") {
            a.device = atoi(value("--device").c_str());
        } else {
            fprintf(stderr, "error: unexpected argument '%s' found\n\n", s.c_str());
            cli_usage(stderr, prog, about, in_help, out_help);
            exit(2);
        }
    }
    if (!have_in || !have_out) {
        fprintf(stderr, "error: the following required arguments were not provided:\n%s%s\n",
                have_in ? "" : "  --input <INPUT>\n", have_out ? "" : "  --output <OUTPUT>\n");
        cli_usage(stderr, prog, about, in_help, out_help);
        exit(2);
    }
    return a;
}
