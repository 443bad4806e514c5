import os,sys
sys.path.insert(0,".")
from ffmpeg_amd import _lib
_lib.select("measure")
sys.argv=["run_case"]+sys.argv[1:]
exec(open("tools/run_case.py").read())
