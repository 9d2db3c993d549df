"""call_mods on the 400,000-row file of r4_parse_prof.sh, in this process (rocprofv3 wants the program itself)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["DSP_NO_SELF_LAUNCH"] = "1"
work = os.environ.get("DSP_WORK", "/tmp/dsp_pipe")
sys.argv = ["deepsignal_plant", "call_mods", "-i", os.path.join(work, "feat_400000.tsv"), "-m", os.path.join(work, "model.ckpt"),
            "-o", os.path.join(work, "calls_prof.tsv"), "-p", "2"]
from deepsignal_plant_amd.deepsignal_plant import main
main()
