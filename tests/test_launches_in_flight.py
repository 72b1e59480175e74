"""scripts/launches_in_flight.py on a hand-made kernel trace: two queues, launches of 60 us started every 30 us."""
import csv
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_overlap_of_two_queues(tmp_path):
    path = tmp_path / "trace.csv"
    fields = ["Kernel_Name", "Queue_Id", "Start_Timestamp", "End_Timestamp", "Grid_Size_X", "Workgroup_Size_X"]
    with open(path, "w", newline="") as fp:
        w = csv.DictWriter(fp, fieldnames=fields)
        w.writeheader()
        for k in range(40):
            w.writerow({"Kernel_Name": "void ss::k_scan_step<0>(ss::StepArgs)", "Queue_Id": 2 + (k & 1), "Start_Timestamp": 30_000 * k,
                        "End_Timestamp": 30_000 * k + 60_000, "Grid_Size_X": 2196 * 512, "Workgroup_Size_X": 512})
        w.writerow({"Kernel_Name": "other_kernel", "Queue_Id": 1, "Start_Timestamp": 0, "End_Timestamp": 5, "Grid_Size_X": 64, "Workgroup_Size_X": 64})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "launches_in_flight.py"), str(path)], capture_output=True, text=True, check=True).stdout
    assert "40 launches" in out and "mean launch duration 60.00 us" in out and "wall time per launch 30.75 us" in out
    assert "launches in flight 1.95" in out and "queue 2: 20, queue 3: 20" in out and "2196: 40, 60.0" in out
