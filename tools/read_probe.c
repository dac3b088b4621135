// read_probe.c - how fast T threads bring a page-cache-resident file into a private buffer, three ways (host only, no GPU):
//   pread    pread(2) into the buffer (what the span loader's reader pool does: host/span_planner.cc)
//   mmap     mmap(MAP_SHARED) of the file, memcpy out of the mapping (page-table entries made by faults, fault-around)
//   mmapnt   the same with non-temporal stores into the buffer (no read-for-ownership of the destination lines)
// Every thread copies whole 8 MB pieces of a `span` (default 256 MB) that moves through the file; per mode the file is gone through
// once, spans in file order, the destination buffer reused (as the pipeline's four span buffers are).  Prints GB/s per mode.
// usage: read_probe FILE threads [span_mb] [numa_node_for_threads|-1]
// build: gcc -O2 -pthread -mavx2 -o read_probe tools/read_probe.c
#define _GNU_SOURCE
#include <fcntl.h>
#include <immintrin.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

static int g_fd, g_threads, g_mode;
static uint8_t *g_buf, *g_map;
static uint64_t g_span_off, g_span_len;
static pthread_barrier_t g_bar;
static volatile int g_stop;
static const uint64_t kPiece = 8ull << 20;

static double now(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + t.tv_nsec * 1e-9;
}

static void copy_nt(uint8_t *dst, const uint8_t *src, uint64_t n) {
    uint64_t i = 0;
    for (; i + 128 <= n; i += 128) {
        __m256i a = _mm256_loadu_si256((const __m256i *)(src + i)), b = _mm256_loadu_si256((const __m256i *)(src + i + 32));
        __m256i c = _mm256_loadu_si256((const __m256i *)(src + i + 64)), d = _mm256_loadu_si256((const __m256i *)(src + i + 96));
        _mm256_stream_si256((__m256i *)(dst + i), a);
        _mm256_stream_si256((__m256i *)(dst + i + 32), b);
        _mm256_stream_si256((__m256i *)(dst + i + 64), c);
        _mm256_stream_si256((__m256i *)(dst + i + 96), d);
    }
    if (i < n) memcpy(dst + i, src + i, n - i);
    _mm_sfence();
}

static void *worker(void *arg) {
    const long id = (long)arg;
    for (;;) {
        pthread_barrier_wait(&g_bar);
        if (g_stop) return NULL;
        const uint64_t n_pieces = (g_span_len + kPiece - 1) / kPiece;
        for (uint64_t p = (uint64_t)id; p < n_pieces; p += (uint64_t)g_threads) {
            const uint64_t off = p * kPiece, len = off + kPiece <= g_span_len ? kPiece : g_span_len - off;
            if (g_mode == 0) {
                uint64_t done = 0;
                while (done < len) {
                    ssize_t g = pread(g_fd, g_buf + off + done, len - done, (off_t)(g_span_off + off + done));
                    if (g <= 0) { perror("pread"); exit(1); }
                    done += (uint64_t)g;
                }
            } else if (g_mode == 1) memcpy(g_buf + off, g_map + g_span_off + off, len);
            else copy_nt(g_buf + off, g_map + g_span_off + off, len);
        }
        pthread_barrier_wait(&g_bar);
    }
}

int main(int argc, char **argv) {
    if (argc < 3) return fprintf(stderr, "usage: read_probe FILE threads [span_mb] [cpu_list_first,last]\n"), 2;
    g_threads = atoi(argv[2]);
    const uint64_t span = (uint64_t)(argc > 3 ? atoi(argv[3]) : 256) << 20;
    g_fd = open(argv[1], O_RDONLY);
    struct stat st;
    if (g_fd < 0 || fstat(g_fd, &st)) return perror("open"), 1;
    const uint64_t size = (uint64_t)st.st_size;
    g_buf = mmap(NULL, span, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    madvise(g_buf, span, MADV_HUGEPAGE);
    memset(g_buf, 1, span);
    pthread_barrier_init(&g_bar, NULL, (unsigned)g_threads + 1);
    pthread_t th[256];
    for (long i = 0; i < g_threads; ++i) pthread_create(&th[i], NULL, worker, (void *)i);
    const char *names[3] = {"pread", "mmap", "mmapnt"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            g_mode = mode;
            const double t0 = now();
            double t_map = 0;
            if (mode) {
                const double a = now();
                g_map = mmap(NULL, size, PROT_READ, MAP_SHARED, g_fd, 0);
                if (g_map == MAP_FAILED) return perror("mmap"), 1;
                madvise(g_map, size, MADV_SEQUENTIAL);
                t_map += now() - a;
            }
            for (uint64_t off = 0; off < size; off += span) {
                g_span_off = off;
                g_span_len = off + span <= size ? span : size - off;
                pthread_barrier_wait(&g_bar);
                pthread_barrier_wait(&g_bar);
            }
            if (mode) {
                const double a = now();
                munmap(g_map, size);
                t_map += now() - a;
            }
            const double dt = now() - t0;
            printf("%-7s %2d threads: %6.2f GB/s (%.3f s for %.1f GB; mmap + munmap %.3f s)\n", names[mode], g_threads, size / 1e9 / dt, dt, size / 1e9, t_map);
            fflush(stdout);
        }
    g_stop = 1;
    pthread_barrier_wait(&g_bar);
    for (long i = 0; i < g_threads; ++i) pthread_join(th[i], NULL);
    return 0;
}
