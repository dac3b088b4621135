// Where does the page cache hold a file?  Maps it, touches one byte per sampled page (which maps the cached page, or reads it in),
// and asks the kernel for the NUMA node of each (move_pages with no target nodes only reports).  gcc -O2 -o pagecache_nodes
// usage: pagecache_nodes FILE [sample every N pages, default 256]
#define _GNU_SOURCE
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const long step = argc > 2 ? atol(argv[2]) : 256;
    int fd = open(argv[1], O_RDONLY);
    if (fd < 0) { perror("open"); return 1; }
    struct stat st;
    fstat(fd, &st);
    const long ps = sysconf(_SC_PAGESIZE);
    const size_t n_pages = (st.st_size + ps - 1) / ps;
    unsigned char *m = mmap(NULL, st.st_size, PROT_READ, MAP_SHARED, fd, 0);
    if (m == MAP_FAILED) { perror("mmap"); return 1; }
    // resident or not, before touching anything
    size_t resident = 0;
    unsigned char *vec = malloc(n_pages);
    if (vec && mincore(m, st.st_size, vec) == 0)
        for (size_t i = 0; i < n_pages; ++i) resident += vec[i] & 1;
    const size_t n = (n_pages + step - 1) / step;
    void **pages = malloc(n * sizeof *pages);
    int *status = malloc(n * sizeof *status);
    volatile unsigned char sink = 0;
    for (size_t i = 0; i < n; ++i) {
        pages[i] = m + i * step * ps;
        sink += *(unsigned char *)pages[i];
    }
    long rc = syscall(SYS_move_pages, 0, (unsigned long)n, pages, NULL, status, 0);
    if (rc < 0) { perror("move_pages"); return 1; }
    long hist[64] = {0}, other = 0;
    for (size_t i = 0; i < n; ++i) {
        if (status[i] >= 0 && status[i] < 64) hist[status[i]]++;
        else other++;
    }
    printf("%s: %.1f MB, %zu of %zu pages resident before the touch (%.1f %%); %zu pages sampled:", argv[1], st.st_size / 1e6, resident, n_pages,
           100.0 * resident / (n_pages ? n_pages : 1), n);
    for (int k = 0; k < 64; ++k)
        if (hist[k]) printf(" node%d %.1f %%", k, 100.0 * hist[k] / n);
    if (other) printf(" other %.1f %%", 100.0 * other / n);
    printf("\n");
    // by tenth of the file: is the placement striped or in runs?
    printf("  by tenth of the file (share on node 0):");
    for (int d = 0; d < 10; ++d) {
        size_t a = n * d / 10, b = n * (d + 1) / 10, z = 0;
        for (size_t i = a; i < b; ++i) z += status[i] == 0;
        printf(" %.0f", 100.0 * z / (b > a ? b - a : 1));
    }
    printf("\n");
    (void)sink;
    return 0;
}
