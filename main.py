"""Entry point kept at the repository root like the reference's main.py: ``python main.py [...] {train,overfit,test}``."""
from vn_pointcloudcompletion_amd.main import main

if __name__ == "__main__":
    main()
