// serve.cc — see serve.h.  One request at a time over a unix stream socket (a session has one caller): a `call`'s arguments
// with absolute paths and the caller's stdout as a passed descriptor; the answer is the exit status and the message the caller
// prints.  The rows go straight from the server into the caller's stdout: byte for byte what `inquistr call` writes itself.
#include "serve.h"

#include "hostapi.h"

#include <poll.h>
#include <signal.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <sys/un.h>
#include <sys/wait.h>
#include <fcntl.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace inq {
namespace {

constexpr uint32_t kMagicCall = 0x31514e49u;  // "INQ1"
constexpr uint32_t kMagicQuit = 0x51514e49u;  // "INQQ"
constexpr uint32_t kMaxBody = 1u << 20;

volatile sig_atomic_t g_stop = 0;
void on_signal(int) { g_stop = 1; }

bool write_all(int fd, const void *p, size_t n) {
    const char *c = (const char *)p;
    while (n) {
        const ssize_t w = ::send(fd, c, n, MSG_NOSIGNAL);
        if (w < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        c += w, n -= (size_t)w;
    }
    return true;
}
bool read_all(int fd, void *p, size_t n) {
    char *c = (char *)p;
    while (n) {
        const ssize_t r = ::recv(fd, c, n, 0);
        if (r < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        if (r == 0) return false;
        c += r, n -= (size_t)r;
    }
    return true;
}

void put_u32(std::vector<char> &b, uint32_t v) { b.insert(b.end(), (char *)&v, (char *)&v + 4); }
void put_u64(std::vector<char> &b, uint64_t v) { b.insert(b.end(), (char *)&v, (char *)&v + 8); }
void put_str(std::vector<char> &b, const char *s) {
    b.push_back(s ? 1 : 0);
    const uint32_t n = s ? (uint32_t)std::strlen(s) : 0u;
    put_u32(b, n);
    if (n) b.insert(b.end(), s, s + n);
}
struct Reader {
    const char *p, *e;
    bool ok = true;
    template <class V>
    V get() {
        V v{};
        if ((size_t)(e - p) < sizeof(V)) ok = false;
        else std::memcpy(&v, p, sizeof(V)), p += sizeof(V);
        return v;
    }
    bool str(std::string &s, bool &present) {
        present = get<char>() != 0;
        const uint32_t n = get<uint32_t>();
        if (!ok || (size_t)(e - p) < n) return ok = false;
        s.assign(p, n), p += n;
        return true;
    }
};

// a path the server can open: the caller's working directory in front of a relative one
std::string absolute(const char *path) {
    if (!path || path[0] == '/') return path ? path : "";
    char cwd[PATH_MAX];
    if (!::getcwd(cwd, sizeof cwd)) return path;
    return std::string(cwd) + "/" + path;
}

bool fill_addr(const char *path, sockaddr_un &sa) {
    std::memset(&sa, 0, sizeof sa);
    sa.sun_family = AF_UNIX;
    if (std::strlen(path) >= sizeof sa.sun_path) return false;
    std::strcpy(sa.sun_path, path);
    return true;
}

int connect_to(const char *path) {
    sockaddr_un sa;
    if (!fill_addr(path, sa)) return -1;
    const int fd = ::socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
    if (fd < 0) return -1;
    if (::connect(fd, (sockaddr *)&sa, sizeof sa) != 0) {
        ::close(fd);
        return -1;
    }
    // only this user's server gets a caller's arguments and its stdout (a socket somebody else put at a guessable path - the
    // INQ_SERVER=auto one under /tmp - is not a server to talk to)
    ucred cred;
    socklen_t len = sizeof cred;
    if (::getsockopt(fd, SOL_SOCKET, SO_PEERCRED, &cred, &len) != 0 || cred.uid != ::getuid()) {
        ::close(fd);
        return -1;
    }
    return fd;
}

// header (magic, body length) with an optional descriptor riding along
bool send_header(int sock, uint32_t magic, uint32_t body_len, int pass_fd) {
    uint32_t hdr[2] = {magic, body_len};
    iovec iov{hdr, sizeof hdr};
    msghdr mh;
    std::memset(&mh, 0, sizeof mh);
    mh.msg_iov = &iov, mh.msg_iovlen = 1;
    alignas(cmsghdr) char ctl[CMSG_SPACE(sizeof(int))];
    if (pass_fd >= 0) {
        std::memset(ctl, 0, sizeof ctl);
        mh.msg_control = ctl, mh.msg_controllen = sizeof ctl;
        cmsghdr *cm = CMSG_FIRSTHDR(&mh);
        cm->cmsg_level = SOL_SOCKET, cm->cmsg_type = SCM_RIGHTS, cm->cmsg_len = CMSG_LEN(sizeof(int));
        std::memcpy(CMSG_DATA(cm), &pass_fd, sizeof(int));
    }
    for (;;) {
        const ssize_t w = ::sendmsg(sock, &mh, MSG_NOSIGNAL);
        if (w < 0 && errno == EINTR) continue;
        return w == (ssize_t)sizeof hdr;
    }
}
bool recv_header(int sock, uint32_t &magic, uint32_t &body_len, int &got_fd) {
    uint32_t hdr[2] = {0, 0};
    iovec iov{hdr, sizeof hdr};
    msghdr mh;
    std::memset(&mh, 0, sizeof mh);
    mh.msg_iov = &iov, mh.msg_iovlen = 1;
    alignas(cmsghdr) char ctl[CMSG_SPACE(sizeof(int))];
    mh.msg_control = ctl, mh.msg_controllen = sizeof ctl;
    got_fd = -1;
    ssize_t r;
    do r = ::recvmsg(sock, &mh, MSG_CMSG_CLOEXEC);
    while (r < 0 && errno == EINTR);
    for (cmsghdr *cm = CMSG_FIRSTHDR(&mh); r >= 0 && cm; cm = CMSG_NXTHDR(&mh, cm))
        if (cm->cmsg_level == SOL_SOCKET && cm->cmsg_type == SCM_RIGHTS && cm->cmsg_len >= CMSG_LEN(sizeof(int))) std::memcpy(&got_fd, CMSG_DATA(cm), sizeof(int));
    if (r != (ssize_t)sizeof hdr) {  // (a stream socket may split 8 bytes in theory; in practice a header arrives whole or not at all)
        if (got_fd >= 0) ::close(got_fd), got_fd = -1;
        return false;
    }
    magic = hdr[0], body_len = hdr[1];
    return true;
}

void answer(int sock, int32_t status, const std::string &msg) {
    const uint32_t n = (uint32_t)msg.size();
    write_all(sock, &status, 4) && write_all(sock, &n, 4) && (n == 0 || write_all(sock, msg.data(), n));
}

}  // namespace

int serve_main(const char *socket_path, int device, double idle_exit_s) {
    sockaddr_un sa;
    if (!socket_path || !fill_addr(socket_path, sa)) {
        std::fprintf(stderr, "serve: socket path missing or too long\n");
        return 2;
    }
    // a stale socket file of a server that is gone may be replaced; a live one may not
    if (int probe = connect_to(socket_path); probe >= 0) {
        ::close(probe);
        std::fprintf(stderr, "serve: %s is in use by a running server\n", socket_path);
        return 1;
    }
    ::unlink(socket_path);
    const int ls = ::socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
    if (ls < 0) return 1;
    const mode_t old = ::umask(0077);  // the socket is the owner's alone: whoever can connect can have files read with the server's rights
    const int brc = ::bind(ls, (sockaddr *)&sa, sizeof sa);
    ::umask(old);
    if (brc != 0 || ::listen(ls, 64) != 0) {
        std::fprintf(stderr, "serve: cannot listen on %s: %s\n", socket_path, std::strerror(errno));
        ::close(ls);
        return 1;
    }
    struct sigaction sg;
    std::memset(&sg, 0, sizeof sg);
    sg.sa_handler = on_signal;
    ::sigaction(SIGTERM, &sg, nullptr);
    ::sigaction(SIGINT, &sg, nullptr);
    ::signal(SIGPIPE, SIG_IGN);  // a caller that is gone (its stdout closed) fails its own call, not the server

    inq_session_t *S = nullptr;
    if (host_api().session_open(device, &S) != 0) {  // returns at once: the HIP runtime starts on a thread of its own
        ::close(ls);
        ::unlink(socket_path);
        return 1;
    }
    std::fprintf(stderr, "serve: listening on %s (device %d)\n", socket_path, device);
    // Two threads: the FRONT one accepts a caller, reads its request and stages the file (opens it, validates the targets, plans
    // the spans, starts reading and uploading them) as soon as the file staged before it has been taken; THIS one takes staged
    // files in order, runs them and answers.  With callers queueing - a workflow manager starts several at once - file k + 1 is
    // staged while file k is called, as inside inq_session_call_many.
    struct Item {
        int cs = -1, out_fd = -1;
        inq_staged_t *staged = nullptr;
        bool quit = false;
    };
    std::mutex mu;
    std::condition_variable cv;
    Item *box = nullptr;  // the one staged item waiting for the runner
    bool front_done = false;
    std::atomic<bool> running{false};
    std::thread front([&] {
        auto last = std::chrono::steady_clock::now();
        for (;;) {
            if (g_stop) break;
            pollfd pf{ls, POLLIN, 0};
            const int pr = ::poll(&pf, 1, 200);
            if (pr <= 0) {
                bool idle;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    idle = box == nullptr && !running.load();
                }
                if (!idle) last = std::chrono::steady_clock::now();  // a call is waiting or running: not idle
                else if (idle_exit_s > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - last).count() > idle_exit_s) break;
                continue;
            }
            const int cs = ::accept4(ls, nullptr, nullptr, SOCK_CLOEXEC);
            if (cs < 0) continue;
            {   // a caller that connects and then says nothing must not hold the queue
                timeval tv{10, 0};
                ::setsockopt(cs, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
            }
            uint32_t magic = 0, body_len = 0;
            int out_fd = -1;
            if (!recv_header(cs, magic, body_len, out_fd)) {
                ::close(cs);
                continue;
            }
            std::unique_ptr<Item> it(new Item());
            it->cs = cs, it->out_fd = out_fd;
            bool bad = false;
            if (magic == kMagicQuit) it->quit = true;
            else {
                std::vector<char> body(body_len <= kMaxBody ? body_len : 0);
                bad = magic != kMagicCall || body_len > kMaxBody || out_fd < 0 || !read_all(cs, body.data(), body.size());
                Reader rd{body.data(), body.data() + body.size()};
                inq_call_args_t a;
                std::memset(&a, 0, sizeof a);
                std::string s[5];
                bool have[5] = {false, false, false, false, false};
                if (!bad) {
                    a.minlen = rd.get<uint32_t>();
                    a.support = rd.get<uint64_t>();
                    a.threads = rd.get<uint64_t>();
                    a.unphased = rd.get<char>() ? 1 : 0;
                    a.device = device;  // the context this server holds
                    for (int k = 0; k < 5 && rd.ok; ++k) rd.str(s[k], have[k]);
                    bad = !rd.ok || !have[0];
                }
                if (bad) {
                    answer(cs, INQ_EXIT_ERROR, "serve: malformed request");
                    ::close(cs);
                    if (out_fd >= 0) ::close(out_fd);
                    continue;
                }
                a.bam = s[0].c_str();
                a.region = have[1] ? s[1].c_str() : nullptr;
                a.region_file = have[2] ? s[2].c_str() : nullptr;
                a.sample_name = have[3] ? s[3].c_str() : nullptr;
                a.reference = have[4] ? s[4].c_str() : nullptr;
                {   // at most one file staged ahead of the one that runs: wait until the runner has taken the one before
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return box == nullptr; });
                }
                if (host_api().session_stage(S, &a, &it->staged) != 0) it->staged = nullptr;  // (the strings are copied by the staging)
            }
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return box == nullptr; });
                box = it.release();
            }
            cv.notify_all();
            last = std::chrono::steady_clock::now();
            if (magic == kMagicQuit) break;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            front_done = true;
        }
        cv.notify_all();
    });
    uint64_t served = 0;
    for (;;) {
        Item *it = nullptr;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return box != nullptr || front_done; });
            it = box;
            box = nullptr;
        }
        cv.notify_all();  // the front thread may stage the next file now
        if (!it) break;   // the front thread is gone and nothing is waiting
        if (it->quit) {
            answer(it->cs, 0, "");
        } else if (!it->staged) {
            answer(it->cs, INQ_EXIT_ERROR, "serve: the file could not be staged");
        } else {
            char err[1024] = {0};
            running.store(true);
            const int rc = host_api().session_run(S, it->staged, it->out_fd, err, sizeof err);
            running.store(false);
            answer(it->cs, rc, err);
            ++served;
        }
        if (it->out_fd >= 0) ::close(it->out_fd);
        ::close(it->cs);
        const bool quit = it->quit;
        delete it;
        if (quit) break;
    }
    g_stop = 1;
    {   // a file the front thread staged after the quit was taken (it cannot: quit ends it) - or is staging right now
        std::unique_lock<std::mutex> lk(mu);
        cv.wait_for(lk, std::chrono::seconds(5), [&] { return front_done; });
    }
    front.join();
    if (box) {  // staged, never run
        if (box->staged) host_api().session_discard(box->staged);
        answer(box->cs, INQ_EXIT_ERROR, "serve: the server is leaving");
        if (box->out_fd >= 0) ::close(box->out_fd);
        ::close(box->cs);
        delete box;
    }
    ::close(ls);
    ::unlink(socket_path);
    std::fprintf(stderr, "serve: leaving after %llu calls\n", (unsigned long long)served);
    std::fflush(nullptr);
    const char *fast = std::getenv("INQ_FAST_EXIT");  // as the other commands: the context is left to the operating system
    if (fast && fast[0] == '1') std::_Exit(0);
    host_api().session_close(S);
    return 0;
}

int client_call(const char *socket_path, const inq_call_args_t *a, int out_fd, int *status, std::string *message) {
    const int sock = connect_to(socket_path);
    if (sock < 0) return 0;
    std::vector<char> body;
    put_u32(body, a->minlen);
    put_u64(body, a->support);
    put_u64(body, a->threads);
    body.push_back(a->unphased ? 1 : 0);
    const std::string bam = absolute(a->bam), bed = absolute(a->region_file), ref = absolute(a->reference);
    put_str(body, bam.c_str());
    put_str(body, a->region);
    put_str(body, a->region_file ? bed.c_str() : nullptr);
    // the sample name the caller's own path gives (the server sees the absolute one: same file stem, src/call.rs:91-100)
    put_str(body, a->sample_name);
    put_str(body, a->reference ? ref.c_str() : nullptr);
    if (!send_header(sock, kMagicCall, (uint32_t)body.size(), out_fd)) {  // nothing has reached the server: the caller may do the call itself
        ::close(sock);
        return 0;
    }
    bool ok = write_all(sock, body.data(), body.size());
    int32_t st = 0;
    uint32_t n = 0;
    ok = ok && read_all(sock, &st, 4) && read_all(sock, &n, 4) && n <= kMaxBody;
    std::string msg(ok ? n : 0, '\0');
    ok = ok && (n == 0 || read_all(sock, &msg[0], n));
    ::close(sock);
    if (!ok) return -1;  // the server went away with our descriptor in its hands: rows may have been written already
    *status = st;
    *message = msg;
    return 1;
}

std::string auto_socket_path(int device) {
    const char *run = std::getenv("XDG_RUNTIME_DIR");
    std::string dir = run && *run ? run : "/tmp";
    return dir + "/inquistr-" + std::to_string((unsigned long)::getuid()) + "-dev" + std::to_string(device) + ".sock";
}

// Is a GPU runtime (or a tool that initialises one: rocprofv3's preloaded library does) mapped into this process?  Read from the
// kernel's own list of the process's mappings, so that it also sees what LD_PRELOAD brought in.  `maps_path` is a test seam.
bool gpu_runtime_mapped(const char *maps_path) {
    FILE *f = std::fopen(maps_path ? maps_path : "/proc/self/maps", "r");
    if (!f) return true;  // cannot tell: behave as if it were (the call then runs in this process)
    char line[4096];
    bool hit = false;
    while (!hit && std::fgets(line, sizeof line, f))
        for (const char *lib : {"libhsa-runtime64", "libamdhip64", "librocprofiler", "librocprof-", "libroctracer", "librocm_smi"})
            if (std::strstr(line, lib)) {
                hit = true;
                break;
            }
    std::fclose(f);
    return hit;
}

bool ensure_server(const char *self_exe, const char *socket_path, int device, double idle_exit_s) {
    if (int probe = connect_to(socket_path); probe >= 0) {
        ::close(probe);
        return true;
    }
    // Starting the server means fork + exec.  That is only safe from a process that has not initialised the GPU: normally true here
    // (the host library is not even loaded when a `call` looks for its server), but NOT when a library preloaded into this process
    // has done it - a profiler's tool library does.  Checked, not assumed: with a GPU runtime mapped the auto-start is refused and
    // the caller runs the call in its own process.
    if (gpu_runtime_mapped(nullptr)) return false;
    // a child of a child, in a session of its own, stdio on /dev/null: nothing of it hangs on the caller's terminal or pipes
    const pid_t pid = ::fork();
    if (pid < 0) return false;
    if (pid == 0) {
        if (::setsid() < 0) ::_exit(1);
        const pid_t p2 = ::fork();
        if (p2 != 0) ::_exit(p2 < 0 ? 1 : 0);
        const int dn = ::open("/dev/null", O_RDWR);
        if (dn >= 0) {
            ::dup2(dn, 0), ::dup2(dn, 1), ::dup2(dn, 2);
            if (dn > 2) ::close(dn);
        }
        const std::string dev = std::to_string(device), idle = std::to_string(idle_exit_s);
        ::execl(self_exe, self_exe, "serve", "--socket", socket_path, "--device", dev.c_str(), "--idle-exit", idle.c_str(), (char *)nullptr);
        ::_exit(127);
    }
    int st = 0;
    while (::waitpid(pid, &st, 0) < 0 && errno == EINTR) {
    }
    for (int i = 0; i < 400; ++i) {  // the socket is there before the runtime starts: a few milliseconds
        if (int probe = connect_to(socket_path); probe >= 0) {
            ::close(probe);
            return true;
        }
        ::usleep(5000);
    }
    return false;
}

bool client_quit(const char *socket_path) {
    const int sock = connect_to(socket_path);
    if (sock < 0) return false;
    int32_t st = 0;
    uint32_t n = 0;
    const bool ok = send_header(sock, kMagicQuit, 0, -1) && read_all(sock, &st, 4) && read_all(sock, &n, 4);
    ::close(sock);
    return ok;
}

}  // namespace inq
