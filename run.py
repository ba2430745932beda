"""python run.py --config config/Taobao-10/deepctr_DN+DR.json  (same surface as the reference's run.py)."""
from mamdr_amd.cli import cli, main  # noqa: F401

if __name__ == "__main__":
    cli()
