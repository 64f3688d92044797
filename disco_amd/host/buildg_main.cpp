/*
 * buildG — MI355X-native drop-in for DISCO's BuildGraph stage executable.
 *
 * Same command line and the same output files as the reference (/root/reference/src/BuildGraph/src/main.cpp:79-150,
 * runDisco.sh:200-245), so runDisco.sh works unmodified when this binary sits next to it:
 *     buildG [-pe f1,f2,...] [-se f1,...] -f <out prefix> -p <disco.cfg> [-t threads] [-m GB] [-w n] [--gpu id]
 * -t sets the host thread count AND the number of <prefix>_<t>_parGraph.txt / _containedReads.txt files, as in the
 * reference. The graph itself is built on the GPU through the C-ABI of libdisco_hip.so (include/disco_hip.h).
 */
#include <omp.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "disco_hip.h"
#include "fastx.h"
#include "writer.h"

using Clock = std::chrono::steady_clock;
static double secs(Clock::time_point a) { return std::chrono::duration<double>(Clock::now() - a).count(); }

static void usage()
{
    std::cerr << "\nUsage: buildG [OPTION]...[PARAM]...\n"
              << "  -pe\tcomma separated paired-end (interleaved) read files, fasta/fastq[.gz]\n"
              << "  -se\tcomma separated single-end read files, fasta/fastq[.gz]\n"
              << "  -f\tprefix of all output files\n"
              << "  -p\tparameter file (MinOverlap4BuildGraph is read from it)\n"
              << "  -t\thost threads = number of partial graph files (default: all cores)\n"
              << "  -m\tmaximum host memory in GB (accepted for compatibility)\n"
              << "  --gpu\tGPU to use (default 0)\n";
}

static std::vector<std::string> split(const std::string &s, char d)
{
    std::vector<std::string> v;
    std::stringstream ss(s);
    std::string item;
    while (std::getline(ss, item, d)) v.push_back(item);
    return v;
}

static std::string trim(const std::string &s)
{
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

/* BG/main.cpp:152-176 : key = value lines; default 30 */
static bool read_min_overlap(const std::string &path, uint32_t &mo)
{
    std::ifstream f(path);
    if (!f.is_open()) return false;
    mo = 30;
    std::string line;
    while (std::getline(f, line)) {
        size_t eq = line.find('=');
        if (eq == std::string::npos) continue;
        auto tok = split(line, '=');
        if (tok.size() < 2) continue;
        if (trim(tok[0]) == "MinOverlap4BuildGraph") mo = (uint32_t)std::stoull(trim(tok[1]), nullptr, 0);
    }
    return true;
}

static int die(const std::string &msg)
{
    std::cout << "\nError: " << msg << std::endl; /* the reference prints and exits 0 (BG/Common.h:64); we exit non-zero */
    return 2;
}

#define DISCO_CALL(ctx, expr)                                                              \
    do {                                                                                   \
        if ((expr) < 0) return die(std::string(#expr " : ") + disco_last_error(ctx));      \
    } while (0)

int main(int argc, char **argv)
{
    std::cout << "Software: Disco Assembler BuildGraph, MI355X-native drop-in (disco_amd)\n";
    auto t_main = Clock::now();
    std::vector<std::string> pe, se;
    std::string prefix, cfg;
    int threads = omp_get_max_threads(), gpu = 0;
    unsigned long long mem_gb = 0, wsize = 0;
    std::cout << "PRINTING ARGUMENTS\n";
    for (int i = 0; i < argc; i++) std::cout << argv[i] << ' ';
    std::cout << std::endl;
    if (argc == 1) {
        usage();
        return 0; /* BG/main.cpp:93-102 */
    }
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> std::string { return (i + 1 < argc) ? argv[++i] : std::string(); };
        if (a == "-pe") for (auto &f : split(next(), ',')) pe.push_back(f);
        else if (a == "-se") for (auto &f : split(next(), ',')) se.push_back(f);
        else if (a == "-f") prefix = next();
        else if (a == "-t") threads = (int)std::stoull(next(), nullptr, 0);
        else if (a == "-w") wsize = std::stoull(next(), nullptr, 0);
        else if (a == "-m") mem_gb = std::stoull(next(), nullptr, 0);
        else if (a == "-p") cfg = next();
        else if (a == "--gpu") gpu = (int)std::stoull(next(), nullptr, 0);
        else {
            usage();
            if (a == "-h" || a == "--help") return 0;
            std::cerr << "Unknown option: " << a << "\n\n";
            return 1; /* BG/main.cpp:133-148 */
        }
    }
    (void)mem_gb;
    (void)wsize;
    if (threads < 1) threads = 1;
    uint32_t min_overlap = 30;
    if (!read_min_overlap(cfg, min_overlap)) {
        std::cerr << "Unable to open parameter file: " << cfg << std::endl;
        return 1; /* BG/main.cpp:157-160 */
    }
    std::cout << "MinOverlap4BuildGraph = " << min_overlap << std::endl;

    bool ccr = false, gc = false;
    disco::read_checkpoint(prefix, ccr, gc);
    if (gc) { /* BG/main.cpp:48-52 */
        std::cout << "Graph already exists. Using previously built graph...\nExiting graph construction." << std::endl;
        return 0;
    }

    /* ---- reads -------------------------------------------------------------------------------------------------- */
    auto t0 = Clock::now();
    disco::ReadSet rs;
    std::string err;
    disco::HostAlloc pinned;
    pinned.alloc = disco_host_alloc;
    pinned.free = disco_host_free;
    if (!disco::load_reads(pe, se, min_overlap, threads, rs, err, pinned)) return die(err);
    for (auto &fr : rs.files) {
        std::cout << "File name: " << fr.name << "\n"
                  << "  " << fr.good << " good reads in current dataset.\n  " << fr.bad << " bad reads in current dataset.\n  "
                  << (fr.good + fr.bad) << " total reads in current dataset.\n";
    }
    std::cout << "Shortest read length in all datasets: " << rs.shortest << "\n Longest read length in all datasets: " << rs.longest << std::endl;
    if (rs.size() == 0) return die("No reads found in the read files provided! Please check if the filename(s) and path(s) are correct.");
    if (!disco::write_read_id_map(prefix, rs, err)) return die(err);
    const double t_parse = secs(t0);
    std::cout << "Function readDataset() finished in " << t_parse << " Seconds." << std::endl;

    /* ---- graph on the GPU ----------------------------------------------------------------------------------------- */
    t0 = Clock::now();
    disco_params prm{min_overlap, 4, 0, 0};
    disco_ctx *ctx = nullptr;
    if (disco_create(gpu, &prm, &ctx) < 0) return die(std::string("disco_create: ") + disco_last_error(nullptr));
    DISCO_CALL(ctx, disco_upload_reads(ctx, rs.packed, rs.stride_words, rs.len.data(), rs.size()));
    const double t_h2d = secs(t0);
    t0 = Clock::now();
    DISCO_CALL(ctx, disco_build_index(ctx));
    DISCO_CALL(ctx, disco_probe(ctx));
    uint64_t n_cont = 0, e_pre = 0, e_out = 0;
    DISCO_CALL(ctx, disco_mark_contained(ctx, &n_cont));
    std::cout << "\n" << (rs.size() - n_cont) << " Non-contained reads. (Keep as is)\n"
              << n_cont << " contained reads. (Need to change their mate-pair information)" << std::endl;
    DISCO_CALL(ctx, disco_build_edges(ctx, &e_pre));
    DISCO_CALL(ctx, disco_transitive_reduce(ctx, &e_out));
    const double t_graph = secs(t0);
    disco_counters cn;
    DISCO_CALL(ctx, disco_get_counters(ctx, &cn));
    std::cout << "Graph construction complete.\n"
              << "  overlaps (pre-reduction) : " << e_pre << "\n  edges after reduction    : " << e_out << "\n  k-mer probes             : " << cn.probes
              << "\n  k-mer hits               : " << cn.kmer_hits << "\n  cap_bind_sites           : " << cn.cap_bind_sites
              << "\n  asymmetric_pairs         : " << cn.asymmetric_pairs << "\n"
              << "Function buildOverlapGraph() [GPU] finished in " << t_graph << " Seconds (" << (t_graph > 0 ? e_pre / t_graph : 0)
              << " overlaps/s); host->device " << t_h2d << " Seconds." << std::endl;
    if (cn.cap_bind_sites || cn.asymmetric_pairs)
        std::cout << "Note: this input is in the order-dependent regime of the reference (edge cap per k-mer reached or overlaps found from one "
                     "side only); the reference's own result varies with its thread count here."
                  << std::endl;

    /* ---- outputs -------------------------------------------------------------------------------------------------- */
    t0 = Clock::now();
    const bool verbose = getenv("DISCO_VERBOSE") != nullptr;
    auto t1 = Clock::now();
    auto lap = [&](const char *what) {
        if (verbose) fprintf(stderr, "[disco host] %-28s %.3f s\n", what, secs(t1));
        t1 = Clock::now();
    };
    std::vector<disco_contained_row> rows(n_cont);
    if (n_cont && disco_fetch_contained(ctx, rows.data(), n_cont) < 0) return die(disco_last_error(ctx));
    lap("fetch contained rows");
    if (!disco::write_contained(prefix, threads, rows, rs, err)) return die(err);
    lap("write contained rows");
    if (!disco::write_checkpoint(prefix, true, false, false, err)) return die(err);
    std::unique_ptr<disco_edge[]> edges(new disco_edge[std::max<uint64_t>(e_out, 1)]); /* 1.8 GB at 45 M edges: not zero-filled first */
    if (e_out && disco_fetch_edges(ctx, edges.get(), e_out) < 0) return die(disco_last_error(ctx));
    lap("fetch edges");
    /* connected components of the reduced graph dealt out to the files: every node has all its edges in one file, which is
     * what lets parsimplify work on the files independently (the reference gets it from its BFS batches) */
    std::unique_ptr<uint16_t[]> edge_file(new uint16_t[std::max<uint64_t>(e_out, 1)]);
    if (e_out && disco_fetch_edge_files(ctx, (uint32_t)threads, edge_file.get(), e_out) < 0) return die(disco_last_error(ctx));
    lap("partition edges into files");
    disco_destroy(ctx);
    lap("release GPU context");
    if (!disco::write_edges(prefix, threads, edges.get(), e_out, rs, threads, err, e_out ? edge_file.get() : nullptr)) return die(err);
    lap("write edges");
    if (!disco::write_checkpoint(prefix, false, true, true, err)) return die(err);
    std::cout << "Function saveParGraphToFile() finished in " << secs(t0) << " Seconds." << std::endl;
    std::cout << "Function main() finished in " << secs(t_main) << " Seconds." << std::endl;
    return 0;
}
