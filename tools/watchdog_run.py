"""run a script under a faulthandler watchdog: dumps every thread's stack and exits if it is still running after N seconds
   python tools/watchdog_run.py 90 bench.py --gpus 2 ..."""
import faulthandler, runpy, sys
n = int(sys.argv[1])
faulthandler.dump_traceback_later(n, exit=True)
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
