#NEXUS
begin trees;
   translate
       1 mars,
       2 saturn,
       3 jupiter,
   (1: 0.184472, 2: 0.027993, 3: 0.045583);
  end;
