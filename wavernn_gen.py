"""Entry point with the reference's script name: ``python wavernn_gen.py --file mel.npy``."""
from tacotronv2_wavernn_chinese_amd.gen import main

if __name__ == "__main__":
    main()
