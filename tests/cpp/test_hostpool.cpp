// Unit test of l3d::HostPool / l3d::on_threads (line3d_amd/csrc/l3d_hostsort.hpp): every thread index runs exactly once per
// region, regions of different widths follow each other on the same persistent workers, a region started inside a region and
// regions started from several threads at once fall back to threads of their own and still run every index, and a forked child
// gets fresh workers.  Built and run by tests/test_host_units.py.
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

#include "../../line3d_amd/csrc/l3d_hostsort.hpp"

static bool region(unsigned nt)
{
    std::vector<std::atomic<int>> hits(nt);
    for (auto& h : hits) h = 0;
    l3d::on_threads(nt, [&](unsigned t) { hits[t].fetch_add(1); });
    for (auto& h : hits) if (h.load() != 1) return false;
    return true;
}

int main()
{
    int bad = 0;
    for (int rep = 0; rep < 200; ++rep)
        for (unsigned nt : { 1u, 2u, 16u, 3u, 9u }) if (!region(nt)) { fprintf(stderr, "region of %u threads: an index did not run exactly once\n", nt); ++bad; }
    {   // nested
        std::atomic<int> inner{ 0 };
        l3d::on_threads(4, [&](unsigned) { l3d::on_threads(3, [&](unsigned) { inner.fetch_add(1); }); });
        if (inner.load() != 12) { fprintf(stderr, "nested regions ran %d of 12 bodies\n", inner.load()); ++bad; }
    }
    {   // concurrent callers
        std::atomic<int> total{ 0 };
        std::vector<std::thread> callers;
        for (int c = 0; c < 6; ++c) callers.emplace_back([&] { for (int r = 0; r < 50; ++r) l3d::on_threads(5, [&](unsigned) { total.fetch_add(1); }); });
        for (auto& t : callers) t.join();
        if (total.load() != 6 * 50 * 5) { fprintf(stderr, "concurrent regions ran %d of %d bodies\n", total.load(), 6 * 50 * 5); ++bad; }
    }
    {   // a forked child has none of the parent's workers
        const pid_t pid = fork();
        if (pid == 0) _exit(region(8) && region(16) ? 0 : 1);
        int status = 0;
        waitpid(pid, &status, 0);
        if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) { fprintf(stderr, "forked child: region failed\n"); ++bad; }
    }
    if (!bad) printf("ok\n");
    return bad ? 1 : 0;
}
