"""Diagnostics: does a captured hipGraph run independent branches concurrently?  One long kernel (a bf16 matmul) on the capture
stream, N tiny kernels on a forked side stream, joined at the end; replay time against the same work captured on one stream."""
import time

import torch


def main():
    dev = torch.device('cuda:0')
    a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    small = [torch.zeros(4096, device=dev) for _ in range(40)]
    side = torch.cuda.Stream()

    def body(forked):
        main_s = torch.cuda.current_stream()
        if forked:
            side.wait_stream(main_s)
            with torch.cuda.stream(side):
                for t in small:
                    t.add_(1.0)
        c = torch.matmul(a, b)
        if forked:
            main_s.wait_stream(side)
        else:
            for t in small:
                t.add_(1.0)
        return c

    for forked in (False, True):
        for _ in range(3):
            body(forked)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body(forked)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        print('graph, %s: %.3f ms per replay' % ('forked side stream' if forked else 'one stream', (time.time() - t0) / 50 * 1e3), flush=True)
        # eager for comparison
        t0 = time.time()
        for _ in range(50):
            body(forked)
        torch.cuda.synchronize()
        print('eager, %s: %.3f ms per pass' % ('forked side stream' if forked else 'one stream', (time.time() - t0) / 50 * 1e3), flush=True)


if __name__ == '__main__':
    main()
