// host_membw.c - what the host's memory gives the threads this process may use: N threads each copy a private 256 MB buffer to
// another (memcpy = what pread from the page cache into a span buffer is made of), best of several rounds; and the same reading only.
// The one-process-per-GPU budget of DESIGN.md 5 is priced with it: a rank at the link's rate asks 3 bytes of host memory traffic per
// compressed byte (page cache -> span buffer: one read + one write; the device's DMA: one read).
//   gcc -O2 -pthread -o host_membw tools/host_membw.c ; ./host_membw [threads] [MB per thread] [rounds]
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static size_t g_bytes;
static int g_rounds;
static pthread_barrier_t g_bar;
static double g_copy_s[64], g_read_s[64];

static double now(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + t.tv_nsec * 1e-9;
}

static void *work(void *arg) {
    const long id = (long)arg;
    uint8_t *a = malloc(g_bytes), *b = malloc(g_bytes);
    memset(a, 1, g_bytes);
    memset(b, 2, g_bytes);
    for (int r = 0; r < g_rounds; ++r) {
        pthread_barrier_wait(&g_bar);
        const double t0 = now();
        memcpy(b, a, g_bytes);
        pthread_barrier_wait(&g_bar);
        const double t1 = now();
        if (id == 0 && (r <= 1 || t1 - t0 < g_copy_s[r > 0])) g_copy_s[r > 0] = t1 - t0;  // [0] = first (cold) round, [1] = best of the rest
        pthread_barrier_wait(&g_bar);
        const double t2 = now();
        uint64_t s = 0;
        for (size_t i = 0; i < g_bytes; i += 8) s += *(const uint64_t *)(a + i);
        if (s == 42) puts("");
        pthread_barrier_wait(&g_bar);
        const double t3 = now();
        if (id == 0 && (r <= 1 || t3 - t2 < g_read_s[r > 0])) g_read_s[r > 0] = t3 - t2;
    }
    free(a);
    free(b);
    return NULL;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 16;
    g_bytes = (size_t)(argc > 2 ? atoi(argv[2]) : 256) << 20;
    g_rounds = argc > 3 ? atoi(argv[3]) : 6;
    if (n < 1 || n > 64) return 1;
    pthread_barrier_init(&g_bar, NULL, (unsigned)n);
    pthread_t th[64];
    for (long i = 0; i < n; ++i) pthread_create(&th[i], NULL, work, (void *)i);
    for (int i = 0; i < n; ++i) pthread_join(th[i], NULL);
    const double total = (double)g_bytes * n / 1e9;
    printf("%d threads x %zu MB: memcpy %.1f GB/s copied = %.1f GB/s of memory traffic (read + write); read only %.1f GB/s\n", n, g_bytes >> 20,
           total / g_copy_s[1], 2 * total / g_copy_s[1], total / g_read_s[1]);
    return 0;
}
