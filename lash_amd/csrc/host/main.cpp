// main.cpp — `lash` command line for the gfx950 build: the flag surface of the reference's clap definition
// (/root/reference/src/main.rs:26-177) on top of liblash_gfx950.so.
//   lash sketch -f LIST [-o sketch] [-k 16] [-t N] [-a hmh|hll|ull] [-p 10] [-s 42]        (main.rs:30-96, 180-279)
//   lash dist   -q PREFIX -r PREFIX [-o dist] [-t N] [-e fgra|ml] [-m 1|0] [--fp32] [--dm]   (main.rs:107-176, 280-617)
// Extras that do not exist upstream: --gpus N / --device D / --devices LIST (which GPUs to use, one worker each), --batch-mb M, --stream-mb M (files
// larger than M MiB are streamed in chunks with on-device accumulation), --hmh-x-low; dist: --device D, --block-rows N,
// --file-order (rows / columns in list-file order instead of the reference's seeded hash-map order)
// (reference rows per GPU call); both: --layout SPEC (or $LASH_LAYOUT): the crate-internal rules as data, see `lash_layout`
// in include/lash_gfx950.h.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/lash_gfx950.h"
#include "dist.hpp"
#include "sketch_files.hpp"

using namespace lashhost;

namespace {

const char *VERSION = "0.1.4";      // main.rs:27 (hard-coded there although the crate is 0.1.6)

struct Args {
    std::map<std::string, std::string> kv;
    std::map<std::string, bool> flags;
};

void usage()
{
    fprintf(stderr,
            "Fast and Memory Efficient (Meta)genome Sketching via HyperLogLog, HyperMinhash and UltraLogLog (MI355X build)\n\n"
            "Usage: lash <COMMAND>\n\nCommands:\n"
            "  sketch  Sketches genomes and serializes them, sketches are compressed\n"
            "  dist    Computes distance between sketches\n\n"
            "sketch options:\n"
            "  -f, --file <file>            One file containing list of FASTA/FASTQ files (.gz/.zstd supported), one per line\n"
            "  -o, --output <output>        Input a prefix/name for your output files [default: sketch]\n"
            "  -k, --kmer <kmer_length>     Length of the kmer [default: 16]\n"
            "  -t, --threads <threads>      Number of threads to use, default to all logical cores\n"
            "  -a, --algorithm <algorithm>  HyperMinHash (hmh), UltraLogLog (ull), or HyperLogLog (hll) [default: hmh]\n"
            "  -p, --precision <precision>  Specifiy precision, for ull and hll only. [default: 10]\n"
            "  -s, --seed <seed>            Random seed [default: 42]\n"
            "      --aa                     Amino acid sketching (k 1-12); the reference carries this flag commented out\n"
            "      --gpus <n> | --device <d> | --devices <d,d,...>  GPUs to use, one worker each [default: device 0]\n"
            "dist options:\n"
            "  -q, --query <prefix>  -r, --reference <prefix>  -o, --output_file <name> [default: dist]\n"
            "  -t, --threads <n>  -e, --estimator <fgra|ml>  -m, --model <1|0>  --fp32  --dm\n"
            "      --file-order   rows and columns in list-file order (default: the reference's hash-map key order)\n"
            "      --hll-bias <file>  HLL++ bias tables (tools/ref_probe/extract_hll_bias.py) [default: $LASH_HLL_BIAS];\n"
            "                     without them hll estimates <= 5 * 2^p are refused\n");
}

bool parse(int argc, char **argv, int first, const std::map<std::string, std::string> &alias,
           const std::vector<std::string> &bool_flags, Args &out, std::string &err)
{
    for (int i = first; i < argc; ++i) {
        std::string a = argv[i];
        std::string key, val;
        bool has_val = false;
        if (a.rfind("--", 0) == 0) {
            size_t eq = a.find('=');
            key = a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
            if (eq != std::string::npos) { val = a.substr(eq + 1); has_val = true; }
        } else if (a.size() >= 2 && a[0] == '-') {
            auto it = alias.find(a.substr(1, 1));
            if (it == alias.end()) { err = "unexpected argument '" + a + "'"; return false; }
            key = it->second;
            if (a.size() > 2) { val = a.substr(2); has_val = true; }
        } else { err = "unexpected argument '" + a + "'"; return false; }
        bool is_flag = false;
        for (auto &f : bool_flags) if (f == key) is_flag = true;
        if (is_flag) { out.flags[key] = true; continue; }
        if (!has_val) {
            if (i + 1 >= argc) { err = "a value is required for '--" + key + "'"; return false; }
            val = argv[++i];
        }
        out.kv[key] = val;
    }
    return true;
}

bool to_u64(const std::string &s, uint64_t &v)
{
    if (s.empty()) return false;
    char *e = nullptr;
    v = strtoull(s.c_str(), &e, 10);
    return e && *e == 0 && s[0] != '-';
}

int cmd_sketch(int argc, char **argv)
{
    Args a;
    std::string err;
    const std::map<std::string, std::string> alias = {{"f", "file"}, {"o", "output"}, {"k", "kmer"}, {"t", "threads"},
                                                      {"a", "algorithm"}, {"p", "precision"}, {"s", "seed"}};
    if (!parse(argc, argv, 2, alias, {"hmh-x-low", "aa"}, a, err)) { fprintf(stderr, "error: %s\n", err.c_str()); return 2; }
    if (!a.kv.count("file")) { fprintf(stderr, "error: the following required arguments were not provided:\n  --file <file>\n"); return 2; }
    SketchOptions opt;
    const std::string output = a.kv.count("output") ? a.kv["output"] : "sketch";
    const std::string alg = a.kv.count("algorithm") ? a.kv["algorithm"] : "hmh";
    uint64_t k = 16, p = 10, seed = 42, threads = std::thread::hardware_concurrency(), gpus = 0, dev = 0, batch_mb = 64,
             stream_mb = 1024;
    if (a.kv.count("kmer") && !to_u64(a.kv["kmer"], k)) { fprintf(stderr, "error: invalid value for --kmer\n"); return 2; }
    if (a.kv.count("precision") && !to_u64(a.kv["precision"], p)) { fprintf(stderr, "error: invalid value for --precision\n"); return 2; }
    if (a.kv.count("seed") && !to_u64(a.kv["seed"], seed)) { fprintf(stderr, "error: invalid value for --seed\n"); return 2; }
    if (a.kv.count("threads") && !to_u64(a.kv["threads"], threads)) { fprintf(stderr, "error: invalid value for --threads\n"); return 2; }
    if (a.kv.count("gpus") && !to_u64(a.kv["gpus"], gpus)) { fprintf(stderr, "error: invalid value for --gpus\n"); return 2; }
    if (a.kv.count("device") && !to_u64(a.kv["device"], dev)) { fprintf(stderr, "error: invalid value for --device\n"); return 2; }
    if (a.kv.count("batch-mb") && !to_u64(a.kv["batch-mb"], batch_mb)) { fprintf(stderr, "error: invalid value for --batch-mb\n"); return 2; }
    if (a.kv.count("stream-mb") && !to_u64(a.kv["stream-mb"], stream_mb)) { fprintf(stderr, "error: invalid value for --stream-mb\n"); return 2; }
    if (alg == "hmh") opt.algo = LASH_HMH;
    else if (alg == "hll") opt.algo = LASH_HLL;
    else if (alg == "ull") opt.algo = LASH_ULL;
    else { fprintf(stderr, "Algorithm must be either hmh, ull, or hll\n"); return 101; }      // main.rs:245 panic
    if (k < 1 || k > 32) { fprintf(stderr, "k-mer length must be 1-32\n"); return 101; }       // utils.rs:501 panic
    const bool amino = a.flags.count("aa") != 0;                                               // main.rs:97-104 (commented out there)
    if (amino && k > 12) { fprintf(stderr, "k-mer length for amino acid must be 1\xe2\x80\x93" "12\n"); return 101; }   // utils.rs:554 panic
    opt.k = (int)k;
    opt.precision = (int)p;
    opt.seed = seed;
    opt.threads = (int)std::max<uint64_t>(1, threads);
    opt.batch_bytes = std::max<uint64_t>(1, batch_mb) << 20;
    opt.stream_bytes = std::max<uint64_t>(1, stream_mb) << 20;
    opt.flags = (a.flags.count("hmh-x-low") ? LASH_F_HMH_X_LOW : 0) | (amino ? LASH_F_AMINO : 0);
    err = layout_from_option(a.kv.count("layout") ? a.kv["layout"] : "", opt.layout);
    if (!err.empty()) { fprintf(stderr, "error: %s\n", err.c_str()); return 2; }
    if (a.kv.count("devices")) {                              // explicit worker list, e.g. 0,1,2,3 (repeats allowed: 0,0 = two workers on GPU 0)
        const std::string &l = a.kv["devices"];
        size_t at = 0;
        while (at <= l.size()) {
            const size_t c = l.find(',', at);
            uint64_t d = 0;
            if (!to_u64(l.substr(at, c == std::string::npos ? std::string::npos : c - at), d)) { fprintf(stderr, "error: invalid value for --devices\n"); return 2; }
            opt.devices.push_back((int)d);
            if (c == std::string::npos) break;
            at = c + 1;
        }
    } else if (gpus > 0) for (uint64_t d = 0; d < gpus; ++d) opt.devices.push_back((int)d);
    else opt.devices.push_back((int)dev);

    std::vector<std::string> files;
    err = read_list_file(a.kv["file"], files);
    if (!err.empty()) { fprintf(stderr, "Error: %s\n", err.c_str()); return 1; }
    SketchStats st;
    err = sketch_files(opt, files, output, &st);
    if (!err.empty()) { fprintf(stderr, "Error: %s\n", err.c_str()); return 1; }
    err = write_parameters_json(output, alg, opt.k, opt.precision, opt.seed, amino);
    if (!err.empty()) { fprintf(stderr, "Error: %s\n", err.c_str()); return 1; }
    fprintf(stderr, "sketched %llu files (%.3f GB of FASTA/FASTQ text) in %.2f s on %zu GPU(s), %llu batches\n",
            (unsigned long long)st.files, st.bytes / 1e9, st.seconds, opt.devices.size(), (unsigned long long)st.batches);
    return 0;
}

int cmd_dist(int argc, char **argv)
{
    Args a;
    std::string err;
    const std::map<std::string, std::string> alias = {{"q", "query"}, {"r", "reference"}, {"o", "output_file"}, {"t", "threads"},
                                                      {"e", "estimator"}, {"m", "model"}};
    if (!parse(argc, argv, 2, alias, {"fp32", "dm", "file-order"}, a, err)) { fprintf(stderr, "error: %s\n", err.c_str()); return 2; }
    if (!a.kv.count("query") || !a.kv.count("reference")) {
        fprintf(stderr, "error: the following required arguments were not provided:\n  --query <query>\n  --reference <reference>\n");
        return 2;
    }
    DistOptions opt;
    opt.query_prefix = a.kv["query"];
    opt.ref_prefix = a.kv["reference"];
    opt.output_file = a.kv.count("output_file") ? a.kv["output_file"] : "dist";
    opt.estimator = a.kv.count("estimator") ? a.kv["estimator"] : "fgra";
    uint64_t model = 1, threads = std::thread::hardware_concurrency(), dev = 0, block_rows = 0;
    if (a.kv.count("block-rows") && !to_u64(a.kv["block-rows"], block_rows)) { fprintf(stderr, "error: invalid value for --block-rows\n"); return 2; }
    opt.block_rows = (uint32_t)std::min<uint64_t>(block_rows, 0xFFFFFFFFull);
    if (a.kv.count("model") && !to_u64(a.kv["model"], model)) { fprintf(stderr, "error: invalid value for --model\n"); return 2; }
    if (a.kv.count("threads") && !to_u64(a.kv["threads"], threads)) { fprintf(stderr, "error: invalid value for --threads\n"); return 2; }
    if (a.kv.count("device") && !to_u64(a.kv["device"], dev)) { fprintf(stderr, "error: invalid value for --device\n"); return 2; }
    opt.model = (int)model;
    opt.threads = (int)std::max<uint64_t>(1, threads);
    opt.fp32 = a.flags.count("fp32") != 0;
    opt.matrix = a.flags.count("dm") != 0;
    opt.file_order = a.flags.count("file-order") != 0;
    if (a.kv.count("hll-bias")) opt.hll_bias_file = a.kv["hll-bias"];
    else if (const char *e = getenv("LASH_HLL_BIAS")) opt.hll_bias_file = e;
    opt.device = (int)dev;
    if (a.kv.count("devices")) {
        const std::string &l = a.kv["devices"];
        size_t at = 0;
        while (at <= l.size()) {
            const size_t c = l.find(',', at);
            uint64_t d = 0;
            if (!to_u64(l.substr(at, c == std::string::npos ? std::string::npos : c - at), d)) { fprintf(stderr, "error: invalid value for --devices\n"); return 2; }
            opt.devices.push_back((int)d);
            if (c == std::string::npos) break;
            at = c + 1;
        }
    }
    err = layout_from_option(a.kv.count("layout") ? a.kv["layout"] : "", opt.layout);
    if (!err.empty()) { fprintf(stderr, "error: %s\n", err.c_str()); return 2; }
    err = run_dist(opt);
    if (!err.empty()) { fprintf(stderr, "Error: %s\n", err.c_str()); return 1; }
    printf("Distances computed.\n");                                                          // main.rs:615
    return 0;
}

}  // namespace

static void epoch_mark(const char *what)
{
    if (!getenv("LASH_CLI_TIMING")) return;
    const double t = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
    fprintf(stderr, "[lash cli] epoch %.3f  %s\n", t, what);
}

int main(int argc, char **argv)
{
    epoch_mark("main entered");
    printf("\n ************** initializing logger *****************\n\n");                   // main.rs:23
    fflush(stdout);
    if (argc < 2) { usage(); return 2; }
    const std::string cmd = argv[1];
    if (cmd == "sketch") { const int rc = cmd_sketch(argc, argv); epoch_mark("main returning"); return rc; }
    if (cmd == "dist") return cmd_dist(argc, argv);
    if (cmd == "--version" || cmd == "-V") { printf("Genome Sketching via HyperLogLog, HyperMinhash and UltraLogLog %s\n", VERSION); return 0; }
    usage();
    return cmd == "--help" || cmd == "-h" || cmd == "help" ? 0 : 2;
}
