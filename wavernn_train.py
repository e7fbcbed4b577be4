"""Entry point with the reference's script name: ``python wavernn_train.py [--hp_file FILE]``."""
from tacotronv2_wavernn_chinese_amd.train import main

if __name__ == "__main__":
    main()
