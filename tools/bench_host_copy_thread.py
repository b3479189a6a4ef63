import threading, time, torch, numpy as np
t = torch.from_numpy(np.random.rand(160,160,160,4).astype(np.float32)).reshape(-1)
pin = torch.empty(t.numel(), dtype=t.dtype).pin_memory()
def work(tag):
    for _ in range(3): pin.copy_(t)
    t0=time.perf_counter()
    for _ in range(10): pin.copy_(t)
    dt=(time.perf_counter()-t0)/10
    print(tag, f"{dt*1e3:.2f} ms per 66 MB copy = {t.numel()*4/dt/1e9:.1f} GB/s, torch threads {torch.get_num_threads()}")
work("main thread")
th=threading.Thread(target=work,args=("worker thread",)); th.start(); th.join()
# while main thread is busy launching (simulated python spin)
stop=False
def spin():
    x=0
    while not stop: x+=1
sp=threading.Thread(target=spin); sp.start()
th=threading.Thread(target=work,args=("worker thread + busy python main",)); th.start(); th.join()
stop=True; sp.join()
