"""DESIGN.md §5 (kernel table), §6 and §9 put together from tools/design_parts/ and the figures of two bench lines (tools/design_table.py):
    python tools/assemble_design.py profiles/r06_final_bench.json
Sections 0-4, 7 and 8 are edited in DESIGN.md itself."""
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
P = os.path.join(ROOT, "tools", "design_parts")


def part(name):
    return open(os.path.join(P, name)).read()


def main():
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()
    i5, i7, i9 = s.index("## 5. Kernels"), s.index("## 7. Multi-GPU"), s.index("## 9. Out of scope / open")
    table = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_table.py"), os.path.join(ROOT, "BENCH_r05.json"), sys.argv[1]],
                           stdout=subprocess.PIPE, text=True, check=True).stdout
    new = s[:i5] + part("design_s5_head.md") + table + part("design_s5_tail.md") + "\n" + part("design_s6.md") + "\n" + s[i7:i9] + part("design_s9.md")
    open(path, "w").write(new)
    print(len(new))


if __name__ == "__main__":
    main()
