"""The build-time guard of the F8 conv kernels' asynchronous loads lives in the package (comfyui-sdmatte_amd/check_async_loads.py: build.py needs it wherever the
package is installed); this is its command-line entry for a tree checkout.  usage: python tools/check_async_loads.py <device .s file> [-v]"""
import importlib.util
import os
import sys

_p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "comfyui-sdmatte_amd", "check_async_loads.py")
_s = importlib.util.spec_from_file_location("sdmatte_check_async_loads", _p)
_m = importlib.util.module_from_spec(_s)
_s.loader.exec_module(_m)
check, kernel_symbols, KERNEL_RE = _m.check, _m.kernel_symbols, _m.KERNEL_RE

if __name__ == "__main__":
    checked, problems = check(open(sys.argv[1]).read(), verbose="-v" in sys.argv)
    for p in problems:
        print("VIOLATION:", p)
    print(f"{checked} asynchronous loads checked, {len(problems)} problem(s)")
    sys.exit(1 if problems else 0)
