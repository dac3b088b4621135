// main.cc — `inquistr call`: the CLI surface of the reference's `call` subcommand (src/main.rs:27-64),
// same flags and defaults.  Other subcommands of the reference are out of scope.
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/inquistr_host.h"
#include "hostapi.h"
#include "serve.h"

static void usage(FILE *f) {
    std::fputs(
        "Call lengths\n\nUsage: inquistr call [OPTIONS] <BAM>\n\nArguments:\n  <BAM>  bam file to call STRs in\n\nOptions:\n"
        "  -r, --region <REGION>            region string to genotype expansion in\n"
        "  -R, --region-file <REGION_FILE>  Bed file with region(s) to genotype expansion(s) in\n"
        "  -m, --minlen <MINLEN>            minimal length of insertion/deletion operation [default: 5]\n"
        "  -s, --support <SUPPORT>          minimal number of supporting reads [default: 3]\n"
        "  -t, --threads <THREADS>          Number of parallel threads to use [default: 1]\n"
        "  -u, --unphased                   If reads have to be considered unphased\n"
        "      --sample-name <SAMPLE_NAME>  sample name to use in output\n"
        "      --reference <REFERENCE>      reference fasta for cram decoding\n"
        "      --device <N>                 HIP device ordinal [default: 0]\n"
        "      --devices <N,N,...>          several HIP devices: the loci are split among them by BAM bytes, same output\n"
        "      --ctx-option <KEY=VALUE>     a device-context option (include/inquistr_hip.h, inq_ctx_set_option), repeatable\n"
        "  -h, --help                       Print help\n",
        f);
}

int main(int argc, char **argv) {
    if (argc >= 2 && std::strcmp(argv[1], "combine") == 0) {  // src/main.rs:65-71: one or more .inq files
        if (argc < 3) {
            std::fputs("error: the following required arguments were not provided:\n  <CALLS>...\n\nUsage: inquistr combine <CALLS>...\n", stderr);
            return 2;
        }
        char err[1024] = {0};
        int rc = inq::host_api().combine(argv + 2, (size_t)(argc - 2), 1, err, sizeof err);
        if (rc != 0) std::fprintf(stderr, rc == INQ_EXIT_PANIC ? "thread 'main' panicked:\n%s\n" : "%s\n", err);
        return rc;
    }
    if (argc >= 2 && std::strcmp(argv[1], "outlier") == 0) {  // src/main.rs:75-99
        inq_outlier_args_t o;
        std::memset(&o, 0, sizeof o);
        o.minsize = 10;
        o.zscore = 3.0f;
        o.method = INQ_OUTLIER_ZSCORE;
        const char *combined = nullptr;
        for (int i = 2; i < argc; ++i) {
            std::string s = argv[i];
            auto eq = s.find('=');
            std::string key = (s.size() > 2 && s[0] == '-' && s[1] == '-' && eq != std::string::npos) ? s.substr(0, eq) : s;
            const char *inl = (key.size() != s.size()) ? argv[i] + eq + 1 : nullptr;
            auto val = [&]() -> const char * {
                if (inl) return inl;
                if (i + 1 >= argc) {
                    std::fprintf(stderr, "error: a value is required for '%s' but none was supplied\n", argv[i]);
                    std::exit(2);
                }
                return argv[++i];
            };
            if (key == "--minsize") o.minsize = (uint32_t)std::strtoul(val(), nullptr, 10);
            else if (key == "-z" || key == "--zscore") o.zscore = std::strtof(val(), nullptr);
            else if (key == "--method") {
                const std::string m = val();
                if (m == "zscore") o.method = INQ_OUTLIER_ZSCORE;
                else if (m == "dbscan") o.method = INQ_OUTLIER_DBSCAN;
                else {
                    std::fprintf(stderr, "error: invalid value '%s' for '--method <METHOD>'\n  [possible values: zscore, dbscan]\n", m.c_str());
                    return 2;
                }
            } else if (key == "-s" || key == "--sample") o.sample = val();
            else if (key == "-S" || key == "--subset") o.subset_file = val();
            else if (key == "--device") o.device = (int32_t)std::strtol(val(), nullptr, 10);
            else if (!s.empty() && s[0] == '-' && s.size() > 1) {
                std::fprintf(stderr, "error: unexpected argument '%s' found\n", s.c_str());
                return 2;
            } else if (!combined) combined = argv[i];
            else {
                std::fprintf(stderr, "error: unexpected argument '%s' found\n", s.c_str());
                return 2;
            }
        }
        if (!combined) {
            std::fputs("error: the following required arguments were not provided:\n  <COMBINED>\n\nUsage: inquistr outlier [OPTIONS] <COMBINED>\n", stderr);
            return 2;
        }
        o.combined = combined;
        char err[1024] = {0};
        ::setenv("INQ_FAST_EXIT", "1", 0);  // this process ends with the command: the device context is left to the operating system
        int rc = inq::host_api().outlier(&o, 1, err, sizeof err);
        if (rc != 0) std::fprintf(stderr, rc == INQ_EXIT_PANIC ? "thread 'main' panicked:\n%s\n" : "%s\n", err);
        std::fflush(nullptr);
        const char *fast = std::getenv("INQ_FAST_EXIT");  // the lines went out through write(2): skip the runtime's tear-down (0.15 - 0.2 s)
        if (fast && fast[0] == '1') std::_Exit(rc);
        return rc;
    }
    if (argc >= 2 && std::strcmp(argv[1], "cohort") == 0) {
        // Not a subcommand of the reference: the loop a user writes around `inquiSTR call` for a cohort (one call per BAM with the
        // same targets, then `combine`), run in ONE process so that the HIP runtime starts once.  Every <out-dir>/<sample>.inq is
        // byte for byte what `inquistr call <BAM> ...` prints.
        inq_call_args_t base;
        std::memset(&base, 0, sizeof base);
        base.minlen = 5, base.support = 3, base.threads = 1;
        std::string out_dir, combined;
        std::vector<std::string> bams;
        for (int i = 2; i < argc; ++i) {
            const std::string k = argv[i];
            auto val = [&]() -> const char * {
                if (i + 1 >= argc) {
                    std::fprintf(stderr, "error: a value is required for '%s' but none was supplied\n", argv[i]);
                    std::exit(2);
                }
                return argv[++i];
            };
            if (k == "-r" || k == "--region") base.region = val();
            else if (k == "-R" || k == "--region-file" || k == "--region_file") base.region_file = val();
            else if (k == "-m" || k == "--minlen") base.minlen = (uint32_t)std::strtoul(val(), nullptr, 10);
            else if (k == "-s" || k == "--support") base.support = std::strtoull(val(), nullptr, 10);
            else if (k == "-t" || k == "--threads") base.threads = std::strtoull(val(), nullptr, 10);
            else if (k == "-u" || k == "--unphased") base.unphased = 1;
            else if (k == "--device") base.device = (int32_t)std::strtol(val(), nullptr, 10);
            else if (k == "-o" || k == "--out-dir") out_dir = val();
            else if (k == "--combined") combined = val();
            else if (k.size() > 1 && k[0] == '-') {
                std::fprintf(stderr, "error: unexpected argument '%s' found\n", k.c_str());
                return 2;
            } else bams.push_back(k);
        }
        if (bams.empty() || out_dir.empty()) {
            std::fputs("Usage: inquistr cohort [-r REGION | -R BED] [-m N] [-s N] [-t N] [-u] [--device N] --out-dir DIR [--combined FILE] <BAM>...\n", stderr);
            return 2;
        }
        ::setenv("INQ_FAST_EXIT", "1", 0);
        inq_session_t *S = nullptr;
        if (inq::host_api().session_open(base.device, &S) != 0) return 1;
        std::vector<inq_call_args_t> args(bams.size(), base);
        std::vector<std::string> outs(bams.size());
        std::vector<int> fds(bams.size(), -1), st(bams.size(), 0);
        for (size_t k = 0; k < bams.size(); ++k) {
            args[k].bam = bams[k].c_str();
            char name[4096];
            inq::host_api().host_sample_name(bams[k].c_str(), name, sizeof name);
            outs[k] = out_dir + "/" + name + ".inq";
            fds[k] = ::open(outs[k].c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
            if (fds[k] < 0) {
                std::fprintf(stderr, "cannot write %s\n", outs[k].c_str());
                return 1;
            }
        }
        char err[2048] = {0};
        int rc = inq::host_api().session_call_many(S, args.data(), args.size(), fds.data(), st.data(), err, sizeof err);
        for (int fd : fds) ::close(fd);
        for (size_t k = 0; k < bams.size(); ++k)
            if (st[k] != 0) std::fprintf(stderr, "%s: exit status %d\n", bams[k].c_str(), st[k]);
        if (rc != 0) std::fprintf(stderr, rc == INQ_EXIT_PANIC ? "thread 'main' panicked:\n%s\n" : "%s\n", err);
        if (rc == 0 && !combined.empty()) {
            int cfd = ::open(combined.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
            std::vector<const char *> files;
            for (auto &o : outs) files.push_back(o.c_str());
            rc = cfd < 0 ? 1 : inq::host_api().combine(files.data(), files.size(), cfd, err, sizeof err);
            if (cfd >= 0) ::close(cfd);
            if (rc != 0) std::fprintf(stderr, "%s\n", err);
        }
        std::fflush(nullptr);
        const char *fast = std::getenv("INQ_FAST_EXIT");
        if (fast && fast[0] == '1') std::_Exit(rc);
        inq::host_api().session_close(S);
        return rc;
    }
    if (argc >= 2 && std::strcmp(argv[1], "serve") == 0) {
        // Not a subcommand of the reference: a process that keeps the device context, so that `inquistr call` - one process per
        // sample, as workflow managers start it - does not pay the HIP runtime's start-up (0.1 - 0.4 s) per file.  `call` hands
        // its arguments and its stdout to the server named by INQ_SERVER=<socket>; without one it runs as always.
        std::string sock;
        int device = 0;
        double idle = 0.0;
        for (int i = 2; i < argc; ++i) {
            const std::string k = argv[i];
            auto val = [&]() -> const char * {
                if (i + 1 >= argc) {
                    std::fprintf(stderr, "error: a value is required for '%s' but none was supplied\n", argv[i]);
                    std::exit(2);
                }
                return argv[++i];
            };
            if (k == "--socket") sock = val();
            else if (k == "--device") device = (int)std::strtol(val(), nullptr, 10);
            else if (k == "--idle-exit") idle = std::strtod(val(), nullptr);
            else if (k == "--quit") {
                if (sock.empty()) sock = std::getenv("INQ_SERVER") ? std::getenv("INQ_SERVER") : "";
                if (sock == "auto") sock = inq::auto_socket_path(device);
                return inq::client_quit(sock.c_str()) ? 0 : 1;
            } else {
                std::fprintf(stderr, "error: unexpected argument '%s' found\n", k.c_str());
                return 2;
            }
        }
        if (sock.empty()) {
            std::fputs("Usage: inquistr serve --socket PATH [--device N] [--idle-exit SECONDS]\n       inquistr serve --socket PATH --quit\n", stderr);
            return 2;
        }
        ::setenv("INQ_FAST_EXIT", "1", 0);
        return inq::serve_main(sock.c_str(), device, idle);
    }
    if (argc < 2 || std::strcmp(argv[1], "call") != 0) {
        if (argc >= 2 && (!std::strcmp(argv[1], "-h") || !std::strcmp(argv[1], "--help"))) {
            std::puts("Tool to genotype STRs from long reads (MI355X build: `call` only)\n\nUsage: inquistr call [OPTIONS] <BAM>");
            return 0;
        }
        std::fputs("error: this build provides the `call`, `combine`, `outlier` (and `cohort`: many `call`s in one process; `serve`: a process that keeps the device context for `call`s with INQ_SERVER set) subcommands only\n\nUsage: inquistr call [OPTIONS] <BAM>\n", stderr);
        return 2;
    }
    if (argc == 2) {  // arg_required_else_help
        usage(stderr);
        return 2;
    }
    inq_call_args_t a;
    std::memset(&a, 0, sizeof a);
    a.minlen = 5;
    a.support = 3;
    a.threads = 1;
    std::string bam;
    std::vector<int32_t> devices;  // --devices: one part of the targets per entry (an ordinal may repeat: N parts on one GPU)
    std::vector<std::pair<std::string, long long>> ctx_options;  // --ctx-option key=value
    auto need = [&](int &i) -> const char * {
        if (i + 1 >= argc) {
            std::fprintf(stderr, "error: a value is required for '%s' but none was supplied\n", argv[i]);
            std::exit(2);
        }
        return argv[++i];
    };
    auto num = [&](const char *s, const char *flag) -> unsigned long long {
        char *e = nullptr;
        if (!*s || *s == '-') {
            std::fprintf(stderr, "error: invalid value '%s' for '%s'\n", s, flag);
            std::exit(2);
        }
        unsigned long long v = std::strtoull(s, &e, 10);
        if (*e) {
            std::fprintf(stderr, "error: invalid value '%s' for '%s'\n", s, flag);
            std::exit(2);
        }
        return v;
    };
    for (int i = 2; i < argc; ++i) {
        std::string s = argv[i];
        auto eq = s.find('=');
        std::string key = (s.size() > 2 && s[0] == '-' && s[1] == '-' && eq != std::string::npos) ? s.substr(0, eq) : s;
        const char *inl = (key.size() != s.size()) ? argv[i] + eq + 1 : nullptr;
        auto val = [&]() { return inl ? inl : need(i); };
        if (key == "-r" || key == "--region") a.region = val();
        else if (key == "-R" || key == "--region-file" || key == "--region_file") a.region_file = val();
        else if (key == "-m" || key == "--minlen") a.minlen = (uint32_t)num(val(), "--minlen");
        else if (key == "-s" || key == "--support") a.support = num(val(), "--support");
        else if (key == "-t" || key == "--threads") a.threads = num(val(), "--threads");
        else if (key == "-u" || key == "--unphased") a.unphased = 1;
        else if (key == "--sample-name" || key == "--sample_name") a.sample_name = val();
        else if (key == "--reference") a.reference = val();
        else if (key == "--device") a.device = (int32_t)num(val(), "--device");
        else if (key == "--devices") {
            const std::string list = val();
            for (size_t b = 0; b <= list.size();) {
                const size_t e = std::min(list.find(',', b), list.size());
                devices.push_back((int32_t)num(list.substr(b, e - b).c_str(), "--devices"));
                b = e + 1;
            }
        }
        else if (s == "--ctx-option" || s.rfind("--ctx-option=", 0) == 0) {
            const std::string kv = s == "--ctx-option" ? std::string(need(i)) : s.substr(13);
            const size_t e2 = kv.find('=');
            char *endp = nullptr;
            const long long v = e2 == std::string::npos ? 0 : std::strtoll(kv.c_str() + e2 + 1, &endp, 10);
            if (e2 == std::string::npos || e2 == 0 || e2 + 1 == kv.size() || (endp && *endp)) {
                std::fprintf(stderr, "error: invalid value '%s' for '--ctx-option <KEY=VALUE>'\n", kv.c_str());
                return 2;
            }
            ctx_options.emplace_back(kv.substr(0, e2), v);
        }
        else if (key == "-h" || key == "--help") { usage(stdout); return 0; }
        else if (!s.empty() && s[0] == '-' && s.size() > 1) {
            std::fprintf(stderr, "error: unexpected argument '%s' found\n", s.c_str());
            return 2;
        } else if (bam.empty()) bam = s;
        else {
            std::fprintf(stderr, "error: unexpected argument '%s' found\n", s.c_str());
            return 2;
        }
    }
    if (bam.empty()) {
        std::fputs("error: the following required arguments were not provided:\n  <BAM>\n", stderr);
        return 2;
    }
    a.bam = bam.c_str();
    for (const auto &kv : ctx_options)  // (a call handed to a server runs on the server's context: its options are the server's)
        if (inq::host_api().ctx_option(kv.first.c_str(), kv.second) != 0) {
            std::fprintf(stderr, "error: invalid value '%s=%lld' for '--ctx-option <KEY=VALUE>': no such option, or value out of range\n", kv.first.c_str(), kv.second);
            return 2;
        }
    if (devices.size() == 1) a.device = devices[0], devices.clear();
    if (!devices.empty()) {
        // one process, one thread + one device context per listed device (host/multi_device.cc); a server holds ONE device
        char err[1024] = {0};
        ::setenv("INQ_FAST_EXIT", "1", 0);
        std::vector<inq_part_stats_t> st(devices.size());
        int rc = inq::host_api().genotype_repeats_devices(&a, devices.data(), devices.size(), 1 /* stdout */, st.data(), err, sizeof err);
        if (rc != 0) std::fprintf(stderr, rc == INQ_EXIT_PANIC ? "thread 'main' panicked:\n%s\n" : "%s\n", err);
        if (std::getenv("INQ_TIMING"))
            for (size_t r = 0; r < st.size(); ++r)
                std::fprintf(stderr,
                             "[inq part] %zu of %zu on device %d: status %d, %llu loci, %llu spans, %.1f MB of BAM read by %d reader threads, rows %.3f s "
                             "(span loop %.3f s = %.2f GB/s, waiting for the loader %.3f s, device calls %.3f s), front end %s\n",
                             r, st.size(), st[r].device, st[r].status, (unsigned long long)st[r].loci, (unsigned long long)st[r].spans,
                             st[r].bam_bytes_read / 1e6, st[r].io_threads, st[r].rows_s, st[r].span_loop_s,
                             st[r].span_loop_s > 0 ? st[r].bam_bytes_read / 1e9 / st[r].span_loop_s : 0.0, st[r].wait_loader_s, st[r].device_calls_s,
                             st[r].front == 2 ? "device" : st[r].front == 1 ? "host" : "-");
        std::fflush(nullptr);
        const char *fast = std::getenv("INQ_FAST_EXIT");
        if (fast && fast[0] == '1') std::_Exit(rc);
        return rc;
    }
    if (const char *server_env = std::getenv("INQ_SERVER"); server_env && *server_env) {
        // INQ_SERVER=auto: this user's server for the device, started (detached; it leaves after INQ_SERVER_IDLE seconds without a
        // call, 120 by default) when none is running - the first call of a batch pays the start-up, the others find the context there
        std::string server_path = server_env;
        if (server_path == "auto") {
            server_path = inq::auto_socket_path(a.device);
            char exe[4096];
            const ssize_t n = ::readlink("/proc/self/exe", exe, sizeof exe - 1);
            const char *idle = std::getenv("INQ_SERVER_IDLE");
            if (n > 0) {
                exe[n] = 0;
                (void)inq::ensure_server(exe, server_path.c_str(), a.device, idle && *idle ? std::strtod(idle, nullptr) : 120.0);
            }
        }
        const char *server = server_path.c_str();
        int st = 0;
        std::string msg;
        const int got = inq::client_call(server, &a, 1 /* stdout */, &st, &msg);
        if (got > 0) {
            if (st != 0) std::fprintf(stderr, st == INQ_EXIT_PANIC ? "thread 'main' panicked:\n%s\n" : "%s\n", msg.c_str());
            return st;
        }
        if (got < 0) {
            std::fprintf(stderr, "the server at %s went away during the call\n", server);
            return 1;
        }
        // no server there: the call runs here, as without INQ_SERVER
    }
    char err[1024] = {0};
    ::setenv("INQ_FAST_EXIT", "1", 0);  // this process ends with the call: see run_device_front
    int rc = inq::host_api().genotype_repeats(&a, 1 /* stdout */, err, sizeof err);
    if (rc != 0) {
        if (rc == INQ_EXIT_PANIC)
            std::fprintf(stderr, "thread 'main' panicked:\n%s\n", err);
        else
            std::fprintf(stderr, "%s\n", err);
    }
    // the rows went out through write(2); nothing is buffered.  Skip the atexit handlers of the HIP runtime
    // (tens of milliseconds of tear-down for a process that is over).
    std::fflush(nullptr);
    const char *fast = std::getenv("INQ_FAST_EXIT");  // INQ_FAST_EXIT=0: leave normally (profilers write their output at exit)
    if (fast && fast[0] == '1') std::_Exit(rc);
    return rc;
}
