"""End-to-end smoke of the fit launcher on one GPU: project/experiments/e1/config.yaml with a YAML-declared dataset."""
import os, sys, tempfile, subprocess, yaml
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_fit_gpu import make_dataset
root = tempfile.mkdtemp()
img, msk = make_dataset(root)
exp = os.path.join(root, "experiments", "e1"); os.makedirs(exp)
yaml.safe_dump({"architecture": "Unet", "backbone": "resnet18", "classes": 1, "activation": "sigmoid", "shape": [128, 128, 3],
                "optimizer": "Adam", "lr": 0.002, "batch": 4, "folds_count": 2, "loss": "binary_crossentropy+1.0*dice_loss",
                "metrics": ["dice"], "primary_metric": "val_dice", "stages": [{"epochs": 2}], "draw_examples": False,
                "fit_with": "simple", "datasets": {"simple": {"input_path": img, "output_path": msk}}}, open(os.path.join(exp, "config.yaml"), "w"))
rc = subprocess.call([sys.executable, "-m", "segmentation_training_pipeline_amd.fit", "--project", root, "--num_gpus", "1", "--gpus_per_net", "1", "--folds", "0"])
print("rc", rc, sorted(os.listdir(exp)), os.path.exists(os.path.join(exp, "weights", "best-0.0.weights")))
