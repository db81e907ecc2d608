// bnz_main.cpp -- `bnzhip`: command-line front end over libbzhip.so with the interface of the
// reference's `bnz` (reference bnz/src/main.rs:32-59 usage, :173-257 argument grammar, :259-285 I/O
// plumbing, :292-309 keep/remove policy, :11-14 exit codes).  SURVEY.md section 8(f) row f1.
// No compression logic lives here: the input is fed to bzh_stream_feed in 16 MiB reads (like
// the reference's BufRead loop, memory stays bounded, pipes and inputs larger than memory work, and reading
// overlaps with the GPU passes) and the stream bytes are written as they become final.
#include <cstdio>
#include <cstdlib>
#include <cerrno>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "../../include/bzhip.h"

namespace {
const int SUCCESS = 0, ERR_ARGS = 1, ERR_FILESYSTEM = 2, ERR_OUTPUT = 3; // bnz/src/main.rs:11-14

const char *TAGLINE = "bnzhip: bzip2 encoder with banzai's output, computed on an AMD MI355X";
const char *VERSION = "version alpha 0.3.1-hip";

[[noreturn]] void die(int code, const std::string &msg)
{
    fprintf(stderr, "%s\n", msg.c_str());
    exit(code);
}

[[noreturn]] void usage(bool full)
{
    fprintf(stderr, "%s\n", TAGLINE);
    if (!full) {
        fprintf(stderr, "   run 'bnzhip --help' for a full list of options\n");
        fprintf(stderr, "   run 'bnzhip --info' for information about this software\n%s\n", VERSION);
        exit(ERR_ARGS);
    }
    fprintf(stderr,
            "\n  usage: bnzhip [options] <input_path>\n\n"
            "  options:\n"
            "     --output <path.bz2>    write the stream to this file\n"
            "     --stdout    or   -c    write the stream to standard out\n"
            "     --keep      or   -k    keep the input file\n"
            "     --remove    or   -r    remove the input file\n\n"
            "     -1 to -9               block size in 100 kB units (default -9)\n"
            "     --fast                 same as -1\n"
            "     --best                 same as -9\n\n"
            "     --verbose   or   -v    accepted for compatibility\n\n"
            "  commands:\n"
            "     --help  --info  --version\n\n"
            "  notes:\n"
            "     '-' as input path reads standard in.  Without --output / --stdout the file\n"
            "     '<input_path>.bz2' is written and the input removed; with an explicit output the\n"
            "     input is kept unless --remove is given.  GPU: $BZHIP_DEVICE (default 0), or\n"
            "     $BZHIP_DEVICES=0,1,2,... to spread the blocks over several GPUs of this node;\n"
            "     BZHIP_HUFFMAN=fixed: 2-6 Huffman tables with refinement (smaller, not banzai's exact bytes).\n\n%s\n",
            VERSION);
    exit(SUCCESS);
}

} // namespace

int main(int argc, char **argv)
{
    if (argc <= 1) usage(false);
    enum { ANY, NOARGS, OUTPATH } expect = ANY;
    std::string in_path, out_path;
    bool have_in = false, in_stdin = false, out_stdout = false, have_out = false;
    int keep = -1, level = 9;
    auto set_input = [&](const std::string &p, bool is_stdin) {
        if (have_in) die(ERR_ARGS, "Only one input may be specified");
        have_in = true;
        in_stdin = is_stdin;
        in_path = p;
    };
    auto set_output = [&](const std::string &p, bool is_stdout) {
        if (have_out && !(out_stdout && is_stdout)) die(ERR_ARGS, "Only one output may be specified");
        have_out = true;
        out_stdout = is_stdout;
        out_path = p;
    };
    for (int k = 1; k < argc; k++) {
        const std::string a = argv[k];
        if (expect == OUTPATH) {
            if (!a.empty() && a[0] == '-') die(ERR_ARGS, "Argument '--output' requires a file path");
            set_output(a, false);
            expect = ANY;
        } else if (expect == ANY && a.rfind("--", 0) == 0) {
            if (a == "--help") usage(true);
            else if (a == "--version") die(SUCCESS, VERSION);
            else if (a == "--info")
                die(SUCCESS, std::string(TAGLINE) +
                                 "\n\nBlocks are suffix-sorted by prefix doubling with radix sorts, move-to-front\n"
                                 "coded and Huffman coded by HIP kernels (libbzhip.so); the stream is bit-identical\n"
                                 "to banzai 0.3.1's.\n\n" + VERSION);
            else if (a == "--verbose") {}
            else if (a == "--keep") keep = 1;
            else if (a == "--remove") keep = 0;
            else if (a == "--fast") level = 1;
            else if (a == "--best") level = 9;
            else if (a == "--output") expect = OUTPATH;
            else if (a == "--stdout") set_output("", true);
            else if (a == "--") expect = NOARGS;
            else die(ERR_ARGS, "Unrecognised argument " + a);
        } else if (expect == ANY && !a.empty() && a[0] == '-') {
            if (a == "-") {
                set_input("", true);
            } else {
                for (size_t c = 1; c < a.size(); c++) {
                    const char f = a[c];
                    if (f == 'c') set_output("", true);
                    else if (f == 'k') keep = 1;
                    else if (f == 'r') keep = 0;
                    else if (f == 'v') {}
                    else if (f >= '1' && f <= '9') level = f - '0';
                    else die(ERR_ARGS, std::string("Flag '") + f + "' is not valid");
                }
            }
        } else {
            set_input(a, false);
        }
    }
    if (!have_in) die(ERR_ARGS, "An input must be specified");

    FILE *inf = in_stdin ? stdin : fopen(in_path.c_str(), "rb");
    if (!inf) die(ERR_FILESYSTEM, "[filesystem error] cannot open " + in_path + ": " + strerror(errno));

    FILE *outf = stdout;
    if (have_out && !out_stdout) {
        outf = fopen(out_path.c_str(), "wb");
        if (!outf) die(ERR_FILESYSTEM, "[filesystem error] cannot create " + out_path + ": " + strerror(errno));
    } else if (!have_out && !in_stdin) {
        const std::string p = in_path + ".bz2";
        outf = fopen(p.c_str(), "wb");
        if (!outf) die(ERR_FILESYSTEM, "[filesystem error] cannot create " + p + ": " + strerror(errno));
    }

    // $BZHIP_DEVICES = "0,1,2,3": the GPUs of this node the blocks are spread over (bzh_create_multi: one host thread and
    // context per device inside the library; the stream is the one a single device writes).  That path holds the whole
    // input in memory; with one device (the default) the input is streamed as below.
    if (const char *dl = getenv("BZHIP_DEVICES")) {
        std::vector<int> devices;
        for (const char *q = dl; *q;) {
            char *e = nullptr;
            const long v = strtol(q, &e, 10);
            if (e == q) break;
            devices.push_back((int)v);
            q = *e == ',' ? e + 1 : e;
            if (*e && *e != ',') break;
        }
        const char *hm = getenv("BZHIP_HUFFMAN");
        if (devices.size() > 1 && !(hm && std::string(hm) == "fixed")) {
            std::vector<uint8_t> data;
            std::unique_ptr<uint8_t[]> buf(new uint8_t[(size_t)16 << 20]);
            for (;;) {
                const size_t k = fread(buf.get(), 1, (size_t)16 << 20, inf);
                data.insert(data.end(), buf.get(), buf.get() + k);
                if (k < ((size_t)16 << 20)) {
                    if (ferror(inf)) die(ERR_OUTPUT, "error during compression: read failed");
                    break;
                }
            }
            bzh_multi *m = nullptr;
            int ms = bzh_create_multi(&m, devices.data(), (int)devices.size(), level);
            if (ms != BZH_OK) die(ERR_OUTPUT, std::string("error during compression: ") + bzh_strerror(ms));
            const size_t n = data.size(), cap = n + n / 4 + (n / 70000 + 4) * 4096 + 65536;
            std::unique_ptr<uint8_t[]> out(new (std::nothrow) uint8_t[cap]);
            size_t got = 0;
            static const uint8_t none = 0;
            ms = out ? bzh_multi_encode(m, n ? data.data() : &none, n, out.get(), cap, &got, nullptr) : BZH_E_NOMEM;
            if (ms != BZH_OK) {
                const std::string msg = std::string("error during compression: ") + bzh_strerror(ms) + ": " + bzh_multi_last_error(m);
                bzh_destroy_multi(m);
                die(ERR_OUTPUT, msg);
            }
            bzh_destroy_multi(m);
            if (got && fwrite(out.get(), 1, got, outf) != got) die(ERR_OUTPUT, "error during compression: write failed");
            if (!in_stdin) fclose(inf);
            if (fflush(outf) != 0) die(ERR_OUTPUT, "error during compression: write failed");
            if (outf != stdout) fclose(outf);
            const bool keep_in = keep >= 0 ? keep == 1 : have_out; // bnz/src/main.rs:292-300
            if (!keep_in && !in_stdin && remove(in_path.c_str()) != 0)
                die(ERR_OUTPUT, "error deleting input file: " + std::string(strerror(errno)));
            return SUCCESS;
        }
    }
    const char *devs = getenv("BZHIP_DEVICE");
    bzh_ctx *ctx = nullptr;
    int st = bzh_create(&ctx, devs ? atoi(devs) : 0, level, 0);
    if (st != BZH_OK) die(ERR_OUTPUT, std::string("error during compression: ") + bzh_strerror(st));
    auto fail = [&](const std::string &what) {
        const std::string msg = "error during compression: " + what;
        bzh_destroy(ctx);
        die(ERR_OUTPUT, msg);
    };
    // not a flag: the reference's option grammar stays as it is.  BZHIP_HUFFMAN=fixed selects the opt-in Huffman mode
    // (2..6 tables, real refinement: smaller files that are no longer bit-identical to banzai's)
    const char *hm = getenv("BZHIP_HUFFMAN");
    if (hm && std::string(hm) == "fixed") {
        st = bzh_set_mode(ctx, BZH_MODE_FIXED);
        if (st != BZH_OK) fail(bzh_strerror(st));
    }
    st = bzh_stream_begin(ctx);
    if (st != BZH_OK) fail(bzh_strerror(st));
    const size_t CHUNK = (size_t)16 << 20;
    // plain arrays, not vectors: the output bound is a worst case of a few hundred megabytes of which a feed fills a
    // fraction -- value-initialising it (and copying it when it grows) cost more than encoding a 100 MB file
    std::unique_ptr<uint8_t[]> in(new uint8_t[CHUNK]), out;
    size_t out_cap = 0;
    for (bool eof = false; !eof;) {
        const size_t k = fread(in.get(), 1, CHUNK, inf);
        if (k < CHUNK) {
            if (ferror(inf)) fail("read failed");
            eof = true;
        }
        const size_t cap = bzh_stream_bound(ctx, k); // depends on what is pending and in flight: ask every time
        if (out_cap < cap) {
            out.reset(); // (nothing in it is still needed)
            out_cap = cap + cap / 2;
            out.reset(new (std::nothrow) uint8_t[out_cap]);
            if (!out) fail("out of memory");
        }
        size_t got = 0;
        st = bzh_stream_feed(ctx, in.get(), k, eof ? 1 : 0, out.get(), out_cap, &got);
        if (st != BZH_OK) fail(std::string(bzh_strerror(st)) + ": " + bzh_last_error(ctx));
        if (got && fwrite(out.get(), 1, got, outf) != got) fail("write failed");
    }
    bzh_destroy(ctx);
    if (!in_stdin) fclose(inf);
    if (fflush(outf) != 0) die(ERR_OUTPUT, "error during compression: write failed");
    if (outf != stdout) fclose(outf);

    const bool keep_input = keep >= 0 ? keep == 1 : have_out; // bnz/src/main.rs:292-300
    if (!keep_input && !in_stdin && remove(in_path.c_str()) != 0)
        die(ERR_OUTPUT, "error deleting input file: " + std::string(strerror(errno)));
    return SUCCESS;
}
